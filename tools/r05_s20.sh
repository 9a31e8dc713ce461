#!/bin/bash
# round-5 session 20: the driver's exact bench command (extras and cpu_baseline included), and --gpus 2 on a one-GPU box (must refuse loudly)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s20; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_exact.log 2> $O/bench_driver_exact.err ) 2> $O/time.txt; echo "driver-exact rc=$?"; grep real $O/time.txt
grep '^{' $O/bench_driver_exact.log | tail -1 > $O/bench_driver_exact.json
python3 bench.py --gpus 2 --steps 1 --warmup 0 > $O/gpus2.out 2> $O/gpus2.err; echo "--gpus 2 on this box: rc=$? stdout bytes $(wc -c < $O/gpus2.out)"; tail -1 $O/gpus2.err
python - <<PY
import json
d=json.load(open("$O/bench_driver_exact.json")); r=d["roofline"]; c=d["config"]
print("value", round(d["value"],1), "n_gpus", d["n_gpus"], "ms_per_step", round(d["ms_per_step"],1), "frac", round(r["frac"],4), "stale", r["traffic_source"]["stale"], "parity stale", c["parity"].get("stale"))
print("cpu_baseline", round(d["cpu_baseline"]["value"],1), d["cpu_baseline"]["cores"], "single_batch", round(c["single_batch"]["converged_trajectories_per_s"]), "sc", round(c["sc_mode"]["terminated_trajectories_per_s"]))
PY
