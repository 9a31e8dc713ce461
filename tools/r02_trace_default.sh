#!/bin/bash
# rocprofv3 kernel trace of the default (two-pool) streaming configuration: per-kernel averages vs the bench line's HIP events.
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02_trace2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 2 --warmup 0 --no-extras --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace rc=$?"
cd $ROOT
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_two_pools.csv; done
rm -rf $OUT/trace
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof_two_pools.json
python - <<PY
import json
d=json.load(open("$OUT/bench_under_rocprof_two_pools.json")); r=d["roofline"]
print("bench under rocprof (two pools): value", round(d["value"],1), "ipm avg_launch_ms", r["avg_launch_ms"], "launches", r["launches"], "discretize avg", d["kernels"]["discretize"]["avg_launch_ms"])
PY
head -4 $OUT/kernel_stats_two_pools.csv | cut -c1-150
