#!/bin/bash
# rocprofv3 kernel statistics of the DEFAULT bench command (batch 8192, 1 warm-up + 2 timed steps), without the extra
# SCvx sub-run and CPU baseline so that every ipm_kernel / discretize_kernel launch in the trace is one of the bench's.
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_default
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --no-cpu-baseline --scvx-batch 0 --mpc-batch 0 > $OUT/trace.log 2>&1
echo "trace rc=$?"
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
cd $ROOT
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-160; done
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print('bench value', d['value'], 'ipm avg_launch_ms', d['roofline']['avg_launch_ms'], 'disc avg_launch_ms', d['kernels']['discretize']['avg_launch_ms'])"
