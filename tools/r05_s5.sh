#!/bin/bash
# round-5 session 5: the persistent streaming kernel on hardware -- rows bitwise against the pool engine, then alternating benches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s5; mkdir -p $O
timeout 300 python tests/tools/engine_equal.py 3000 1024 > $O/engine_equal.log 2>&1; echo "engine_equal rc=$?"; tail -8 $O/engine_equal.log
for rep in 1 2; do
for E in 0 1; do
  SCPP_STREAM_ENGINE=$E timeout 400 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_e${E}_$rep.json 2> $O/bench_e${E}_$rep.err; echo "bench engine $E rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_e${E}_$rep.json")); c=d["config"]; r=d["roofline"]
    print("engine $E rep $rep", round(d["value"],1), "conv", c["converged_fraction"], "ipm/traj", round(c["mean_ipm_iterations_per_trajectory"],2), "frac", round(r["frac"],4))
except Exception as e: print("engine $E failed", e, open("$O/bench_e${E}_$rep.err").read()[-800:])
PY
done
done
