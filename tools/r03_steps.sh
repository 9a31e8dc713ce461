#!/bin/bash
# same-box A/B of the number of RKF78 steps per segment in discretize_kernel (compile-time variants)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03_steps; mkdir -p $O
for rep in 1 2; do
  for v in 5 3 2; do
    LIB=scpp_amd/libscpp_hip.so; [ $v != 5 ] && LIB=build/libscpp_steps$v.so
    timeout 300 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --library $LIB > $O/bench_s${v}_$rep.json 2> $O/bench_s${v}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_s${v}_$rep.json")); k=d["kernels"]["discretize"]
    print("steps $v rep $rep", round(d["value"],1), "conv", d["config"]["converged_fraction"], "iters", round(d["config"]["mean_scvx_iterations"],3), "solves", round(d["config"]["mean_subproblem_solves"],3), "disc ms/launch", round(k["avg_launch_ms"],2), "disc kernel_time_s", round(k["kernel_time_s"],3), "ipm span", round(d["roofline"]["avg_launch_ms"],2))
except Exception as e: print("steps $v rep $rep failed", e)
PY
  done
done
