#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03_pools3; mkdir -p $O
run () { # batch pools rep
  timeout 300 python bench.py --batch $1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline --pools $2 > $O/bench_b$1_p$2_$3.json 2> $O/bench_b$1_p$2_$3.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_b$1_p$2_$3.json")); print("batch $1 pools $2 rep $3", round(d["value"],1), "ipm span ms", round(d["roofline"]["avg_launch_ms"],2), d["config"]["rounds"])
except Exception as e: print("batch $1 pools $2 rep $3 failed", e)
PY
}
for rep in 1 2; do for P in 6 7 3; do run 8192 $P $rep; done; done
for P in 1 2 3; do run 4096 $P 1; done
