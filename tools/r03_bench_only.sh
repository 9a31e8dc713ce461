#!/bin/bash
# official bench lines of the round-3 final library (default pools heuristic = 3 pools at 8192 slots)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${OUTDIR:-r03_v2}; mkdir -p $O
timeout -k 5 900 python bench.py > $O/bench_default.log 2>&1; echo "bench default rc=$?"
grep '^{' $O/bench_default.log | tail -1 > $O/bench_default.json
timeout -k 5 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_driver.log 2>&1; echo "bench driver-style rc=$?"
grep '^{' $O/bench_driver.log | tail -1 > $O/bench_driver.json
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "stream or discretize_variants" > $O/pytest_stream.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_stream.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --steps 2 --warmup 0 --no-extras --no-cpu-baseline > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_default_pools.csv; done
rm -rf $O/trace
grep '^{' $O/trace.log | tail -1 > $O/bench_under_rocprof_default_pools.json
python - <<PY
import json
for n in ("bench_default","bench_driver","bench_under_rocprof_default_pools"):
    try:
        d=json.load(open("$O/%s.json"%n)); r=d["roofline"]
        print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "frac", round(r["frac"],4), "span-sum frac", round(r["frac_of_span_sum"],4), "avg_launch_ms", round(r["avg_launch_ms"],3), "launches", r["launches"], "pools", d["config"]["rounds"], "kernel_time_s", round(r["kernel_time_s"],3), "region", round(r["timed_region_s"],3))
    except Exception as e: print(n, "failed", e)
PY
head -4 $O/kernel_stats_default_pools.csv | cut -c1-150
