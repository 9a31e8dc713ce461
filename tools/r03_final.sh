#!/bin/bash
# round-3 final measurement session: GPU parity tests, smoke, default bench line, driver-style bench (--steps 20 --warmup 5),
# rocprofv3 kernel-trace statistics (one pool = un-overlapped, two pools = default), PMC summary (tools/pmc_hbm.sh), pools A/B,
# the C++ RCCL driver.   usage: bash tools/r03_final.sh <tag>     (write the commit id to tools/commit_id.txt before gpurun)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-final}; ROOT=$PWD; OUT=$ROOT/gpurun_out/r03_$TAG; mkdir -p $OUT
export GRAFT_COMMIT=$(cat tools/commit_id.txt 2>/dev/null || echo worktree)
timeout -k 5 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
timeout -k 5 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench default rc=$?"
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
timeout -k 5 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver.log 2>&1; echo "bench driver-style rc=$?"
grep '^{' $OUT/bench_driver.log | tail -1 > $OUT/bench_driver.json
for P in 3 2; do
  timeout -k 5 300 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --pools $P > $OUT/bench_pools$P.log 2>&1
  grep '^{' $OUT/bench_pools$P.log | tail -1 > $OUT/bench_pools$P.json
done
(cd scpp_amd/host && timeout 300 ./scvx_multi_gpu --batch 4096 --gpus 1 --slots 4096 --config ../config > $OUT/scvx_multi_gpu.log 2>&1; echo "scvx_multi_gpu rc=$?"; cat $OUT/scvx_multi_gpu.log)
cd /tmp && export TMPDIR=/tmp
for PP in 1 2; do
  timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$PP -- python $ROOT/bench.py --steps 2 --warmup 0 --pools $PP --no-extras --no-cpu-baseline > $OUT/trace$PP.log 2>&1
  echo "trace pools=$PP rc=$?"
  for f in $(find $OUT/trace$PP -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_pools$PP.csv; done
  rm -rf $OUT/trace$PP
  grep '^{' $OUT/trace$PP.log | tail -1 > $OUT/bench_under_rocprof_pools$PP.json
done
cd $ROOT
bash tools/pmc_hbm.sh r03_$TAG 4096 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_r03_$TAG/summary.json $OUT/pmc_summary.json
python - <<PY
import json
for n in ("bench_default","bench_driver","bench_pools2","bench_pools3","bench_under_rocprof_pools1","bench_under_rocprof_pools2"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); r=d["roofline"]
        print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "frac", round(r["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],3), "launches", r["launches"], "traffic", r["traffic"])
    except Exception as e: print(n, "failed", e)
try:
    d=json.load(open("$OUT/bench_default.json")); c=d["config"]
    print(json.dumps({k:c[k] for k in ("single_pool","single_batch","sc_mode","mpc_mode")},indent=0)[:3000]); print(json.dumps(d.get("cpu_baseline"))[:1800])
    p=json.load(open("$OUT/pmc_summary.json")); print("PMC", p.get("ipm_bytes_per_instance_iteration"), p.get("calibration"), p.get("ipm_l2_hit_rate"), p.get("commit"))
except Exception as e: print("summary failed", e)
PY
head -8 $OUT/kernel_stats_pools1.csv | cut -c1-170
head -8 $OUT/kernel_stats_pools2.csv | cut -c1-170
