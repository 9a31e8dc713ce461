#!/bin/bash
# round 3, GPU session 2: A/B of the elimination's pivot broadcast (INVCHOL_MAXFIRST), full GPU test suite with the new tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03s2; mkdir -p $O
for rep in 1 2; do
  for v in base maxfirst; do
    LIB=scpp_amd/libscpp_hip.so; [ $v = maxfirst ] && LIB=build/libscpp_maxfirst.so
    timeout 300 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --library $LIB > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${v}_$rep.json")); print("$v $rep", round(d["value"],1), "frac", round(d["roofline"]["frac"],4), "ipm ms/launch", round(d["roofline"]["avg_launch_ms"],2), "conv", d["config"]["converged_fraction"])
except Exception as e: print("$v $rep failed", e)
PY
  done
done
timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"
grep -v "^\s*$" $O/pytest_gpu.log | tail -45 | cut -c1-600
