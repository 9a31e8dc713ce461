#!/bin/bash
# round-5 session 2: the split schedule on hardware -- bitwise against the resident kernel, then alternating same-box benches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s2; mkdir -p $O
timeout 600 python tests/tools/split_equal.py scpp_amd/libscpp_hip.so 1024 > $O/split_equal.log 2>&1; echo "split_equal rc=$?"; tail -6 $O/split_equal.log
for rep in 1 2; do
for S in 0 1; do
  SCPP_IPM_SCHEDULE=$S timeout 400 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_s${S}_$rep.json 2> $O/bench_s${S}_$rep.err; echo "bench schedule $S rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_s${S}_$rep.json")); c=d["config"]; r=d["roofline"]
    print("schedule $S rep $rep", round(d["value"],1), "conv", c["converged_fraction"], "ipm/traj", round(c["mean_ipm_iterations_per_trajectory"],2), "ipm span", round(r["avg_launch_ms"],2), "disc ms", round(d["kernels"]["discretize"]["avg_launch_ms"],2))
except Exception as e: print("schedule $S failed", e, open("$O/bench_s${S}_$rep.err").read()[-600:])
PY
done
done
