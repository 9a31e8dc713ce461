#!/bin/bash
# round 3, GPU session 1: permlane probe, A/B of the elimination exchange (permlane vs LDS), full GPU test suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03s1
O=gpurun_out/r03s1
./tools/bin/permlane_probe > $O/probe.txt 2>&1; PROBE=$?
cat $O/probe.txt | head -3
for rep in 1 2; do
  for v in perm lds; do
    LIB=scpp_amd/libscpp_hip.so; [ $v = lds ] && LIB=build/libscpp_lds.so
    if [ $v = perm ] && [ $PROBE -ne 0 ]; then continue; fi
    timeout 300 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --library $LIB > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${v}_$rep.json")); print("$v $rep", round(d["value"],1), "frac", round(d["roofline"]["frac"],4), "ipm ms/launch", round(d["roofline"]["avg_launch_ms"],2), "conv", d["config"]["converged_fraction"])
except Exception as e: print("$v $rep failed", e)
PY
  done
done
if [ $PROBE -ne 0 ]; then export SCPP_HIP_LIBRARY=$PWD/build/libscpp_lds.so; echo "PROBE FAILED: tests run against the LDS variant"; fi
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
