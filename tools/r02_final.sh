#!/bin/bash
# round-2 final measurement session: GPU parity tests, default bench line, driver-style bench (--steps 20 --warmup 5), kernel-trace
# statistics, PMC summary (tools/pmc_hbm.sh).   usage: bash tools/r02_final.sh <tag>
TAG=${1:-final}; ROOT=$PWD; OUT=$ROOT/gpurun_out/r02_$TAG; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout -k 5 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench default rc=$?"
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
timeout -k 5 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver.log 2>&1; echo "bench driver-style rc=$?"
grep '^{' $OUT/bench_driver.log | tail -1 > $OUT/bench_driver.json
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 2 --warmup 0 --pools 1 --no-extras --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace rc=$?"
cd $ROOT
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_single_pool.csv; done
rm -rf $OUT/trace
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof_single_pool.json
bash tools/pmc_hbm.sh $TAG 4096 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/summary.json $OUT/pmc_summary.json
python - <<PY
import json
for n in ("bench_default","bench_driver"):
    d=json.load(open("$OUT/%s.json"%n)); print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"])
d=json.load(open("$OUT/bench_default.json")); c=d["config"]
print(json.dumps({k:c[k] for k in ("single_pool","single_batch","sc_mode")},indent=0)[:2500]); print(json.dumps(d.get("cpu_baseline"))[:1500])
PY
head -8 $OUT/kernel_stats_single_pool.csv | cut -c1-160
