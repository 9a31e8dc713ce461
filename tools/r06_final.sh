#!/bin/bash
# round-6 closing measurement session for ONE library state (profiles/ must reproduce HEAD): GPU parity suite, smoke, rocprofv3 kernel-trace statistics
# (default = persistent engine; pool engine with one pool = the separate kernels un-overlapped), PMC summary (tools/pmc_hbm.sh, incl. the kernel-source
# hash), bench lines (default, driver-style, pool engine driver-style), the -DRES_PREVLANE_MEMORY=1 library against the shipped one, the C++ RCCL driver.
#   usage: bash tools/r06_final.sh <tag>     (write the commit id to tools/commit_id.txt before gpurun)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-final}; ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_$TAG; mkdir -p $OUT
export GRAFT_COMMIT=$(cat tools/commit_id.txt 2>/dev/null || echo worktree)
python tools/csrc_hash.py > $OUT/csrc_sha.txt
timeout -k 5 2700 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
cp gpurun_out/r06_parity_at_scale.json $OUT/parity_at_scale.json 2>/dev/null
cp $OUT/parity_at_scale.json profiles/r06_parity_at_scale.json 2>/dev/null  # the bench lines below import it (config.parity): this session's, not an older one's
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
[ -f build/pre_registry.so ] && { timeout 300 python tests/tools/lib_equal.py scpp_amd/libscpp_hip.so build/pre_registry.so > $OUT/pre_registry_equal.log 2>&1; echo "pre_registry rc=$?"; tail -2 $OUT/pre_registry_equal.log; }
[ -f build/prevlane_memory.so ] && { timeout 300 python tests/tools/lib_equal.py scpp_amd/libscpp_hip.so build/prevlane_memory.so > $OUT/prevlane_equal.log 2>&1; echo "prevlane rc=$?"; tail -2 $OUT/prevlane_equal.log; }
cd /tmp && export TMPDIR=/tmp
for V in persistent pools1; do
  EXTRA=""; [ $V = pools1 ] && EXTRA="--pools 1"
  timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$V -- python $ROOT/bench.py --steps 2 --warmup 0 $EXTRA --no-extras --no-cpu-baseline > $OUT/trace_$V.log 2>&1
  echo "trace $V rc=$?"
  for f in $(find $OUT/trace_$V -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_$V.csv; done
  rm -rf $OUT/trace_$V
  grep '^{' $OUT/trace_$V.log | tail -1 > $OUT/bench_under_rocprof_$V.json
done
cd $ROOT
bash tools/pmc_hbm.sh r06_$TAG 8192 2 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_r06_$TAG/summary.json $OUT/pmc_summary.json
# the bench lines LAST, with this session's PMC summary in place (same kernel-source hash -> not stale)
cp $OUT/pmc_summary.json profiles/r06_pmc_hbm_v9_session.json
timeout -k 5 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench default rc=$?"
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
timeout -k 5 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver.log 2>&1; echo "bench driver-style rc=$?"
grep '^{' $OUT/bench_driver.log | tail -1 > $OUT/bench_driver.json
SCPP_STREAM_ENGINE=0 timeout -k 5 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver_pools.log 2>&1; echo "bench driver-style (pool engine) rc=$?"
grep '^{' $OUT/bench_driver_pools.log | tail -1 > $OUT/bench_driver_pools.json
rm -f profiles/r06_pmc_hbm_v9_session.json
(cd scpp_amd/host && timeout 300 ./scvx_multi_gpu --batch 4096 --gpus 1 --slots 4096 --config ../config > $OUT/scvx_multi_gpu.log 2>&1; echo "scvx_multi_gpu rc=$?"; tail -3 $OUT/scvx_multi_gpu.log | cut -c1-300)
python - <<PY
import json
for n in ("bench_default","bench_driver","bench_driver_pools","bench_under_rocprof_persistent","bench_under_rocprof_pools1"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); r=d["roofline"]
        print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "frac", round(r["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],3), "launches", r["launches"], "traffic", r["traffic"], "stale", (r.get("traffic_source") or {}).get("stale"))
    except Exception as e: print(n, "failed", e)
try:
    d=json.load(open("$OUT/bench_default.json")); c=d["config"]
    print(json.dumps(d["roofline"].get("steps"),indent=0)[:1500])
    print(json.dumps({k:c.get(k) for k in ("single_pool","single_batch","step_length_rule","sc_mode","mpc_mode","parity","what_converged_means")},indent=0)[:5000]); print(json.dumps(d.get("cpu_baseline"))[:1500])
    p=json.load(open("$OUT/pmc_summary.json")); print("PMC", p.get("engine"), p.get("ipm_bytes_per_instance_iteration"), p.get("bytes_per_trajectory"), p.get("calibration"), p.get("ipm_l2_hit_rate"), p.get("commit"), p.get("csrc_sha"))
except Exception as e: print("summary failed", e)
PY
head -6 $OUT/kernel_stats_persistent.csv | cut -c1-200
head -8 $OUT/kernel_stats_pools1.csv | cut -c1-200
