#!/bin/bash
# same-box alternating A/B of build/<name>.so variants.   usage: bash tools/r04_ab.sh <tag> <variant> [<variant> ...]   (first = reference)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=$1; shift; O=gpurun_out/r04_ab_$TAG; mkdir -p $O
REPS=${REPS:-2}; STEPS=${STEPS:-8}
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    timeout 300 python bench.py --steps $STEPS --warmup 2 --no-extras --no-cpu-baseline --library build/$v.so > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${v}_$rep.json")); k=d["kernels"]["discretize"]; c=d["config"]
    print("$v rep $rep", round(d["value"],1), "conv", c["converged_fraction"], "iters", round(c["mean_scvx_iterations"],3), "solves", round(c["mean_subproblem_solves"],3), "ipm/traj", round(c["mean_ipm_iterations_per_trajectory"],2), "disc ms", round(k["avg_launch_ms"],2), "ipm span", round(d["roofline"]["avg_launch_ms"],2), "frac", round(d["roofline"]["frac"],4))
except Exception as e: print("$v rep $rep failed", e, open("$O/bench_${v}_$rep.err").read()[-400:])
PY
  done
done
