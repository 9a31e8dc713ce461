#!/bin/bash
# same-box A/B of the number of slot pools (HIP streams) of the streaming engine at the bench configuration
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${OUTDIR:-r03_pools}; mkdir -p $O
for rep in ${REPS:-1 2 3}; do
  for P in ${POOLS_LIST:-2 3 4}; do
    timeout 300 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --pools $P > $O/bench_p${P}_$rep.json 2> $O/bench_p${P}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_p${P}_$rep.json")); print("pools $P rep $rep", round(d["value"],1), "ipm span ms", round(d["roofline"]["avg_launch_ms"],2), "excl ms/launch", round(d["roofline"]["exclusive_ms_per_launch"],2), "rounds", d["config"]["rounds"])
except Exception as e: print("pools $P rep $rep failed", e)
PY
  done
done
