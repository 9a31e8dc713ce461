#!/bin/bash
# round-3 session after the adaptive RKF78 step count: GPU parity tests, smoke, default bench line, driver-style bench, rocprofv3
# kernel statistics of the default configuration.    usage: bash tools/r03_v4.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-v4}; ROOT=$PWD; OUT=$ROOT/gpurun_out/r03_$TAG; mkdir -p $OUT
export GRAFT_COMMIT=$(cat tools/commit_id.txt 2>/dev/null || echo worktree)
timeout -k 5 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
timeout -k 5 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench default rc=$?"
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
timeout -k 5 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver.log 2>&1; echo "bench driver-style rc=$?"
grep '^{' $OUT/bench_driver.log | tail -1 > $OUT/bench_driver.json
(cd scpp_amd/host && timeout 300 ./scvx_multi_gpu --batch 4096 --gpus 1 --slots 4096 --config ../config > $OUT/scvx_multi_gpu.log 2>&1; echo "scvx_multi_gpu rc=$?"; tail -4 $OUT/scvx_multi_gpu.log)
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 2 --warmup 0 --no-extras --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace rc=$?"
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_default.csv; done
rm -rf $OUT/trace
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
cd $ROOT
python - <<PY
import json
for n in ("bench_default","bench_driver","bench_under_rocprof"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); r=d["roofline"]
        print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "frac", round(r["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],3), "launches", r["launches"], "disc", json.dumps(d["kernels"]["discretize"])[:600])
    except Exception as e: print(n, "failed", e)
PY
head -8 $OUT/kernel_stats_default.csv
