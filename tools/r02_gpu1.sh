#!/bin/bash
# round-2 GPU session 1: parity tests, the default bench line, slot-pool sweep, kernel-trace statistics
ROOT=$PWD; OUT=$ROOT/gpurun_out/r02a; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 5 600 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?"
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
for P in 1 2 4; do
  timeout -k 5 300 python bench.py --pools $P --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/bench_pools$P.log 2>&1
  grep '^{' $OUT/bench_pools$P.log | tail -1 > $OUT/bench_pools$P.json
  python -c "
import json; d=json.load(open('$OUT/bench_pools$P.json')); print('pools $P value', d['value'], 'ms/step', d['ms_per_step'], 'ipm avg ms', d['roofline']['avg_launch_ms'], d['config']['rounds'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace rc=$?"
cd $ROOT
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-200; cp $f $OUT/kernel_stats.csv; done
rm -rf $OUT/trace
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(json.dumps({k:d[k] for k in ('value','ms_per_step')})); print(json.dumps(d['config'],indent=0)[:3000]); print(json.dumps(d['roofline'])[:1500]); print(json.dumps(d.get('cpu_baseline'))[:1500])"
