#!/bin/bash
# same-box alternating A/B of the shipped library against build/<name>.so variants.   usage: bash tools/r03_ab_lib.sh <tag> <variant> [<variant> ...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=$1; shift; O=gpurun_out/r03_ab_$TAG; mkdir -p $O
for rep in 1 2; do
  for v in base "$@"; do
    LIB=scpp_amd/libscpp_hip.so; [ $v != base ] && LIB=build/$v.so
    timeout 300 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --library $LIB > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${v}_$rep.json")); k=d["kernels"]["discretize"]
    print("$v rep $rep", round(d["value"],1), "conv", d["config"]["converged_fraction"], "iters", round(d["config"]["mean_scvx_iterations"],3), "solves", round(d["config"]["mean_subproblem_solves"],3), "disc ms/launch", round(k["avg_launch_ms"],2), "ipm span", round(d["roofline"]["avg_launch_ms"],2), "ipm union s", round(d["roofline"]["kernel_time_s"],3), "timed", round(d["roofline"]["timed_region_s"],3))
except Exception as e: print("$v rep $rep failed", e)
PY
  done
done
