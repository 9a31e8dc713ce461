#!/bin/bash
# round-4 measurement session for ONE library state (profiles/ must reproduce HEAD): GPU parity tests, smoke, default bench
# line, driver-style bench, step-length-rule bench, rocprofv3 kernel-trace statistics (one pool = un-overlapped; default pools), PMC
# summary (tools/pmc_hbm.sh, incl. the kernel-source hash), SQ counters, the C++ RCCL driver.
#   usage: bash tools/r04_final.sh <tag>     (write the commit id to tools/commit_id.txt before gpurun)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-final}; ROOT=$PWD; OUT=$ROOT/gpurun_out/r04_$TAG; mkdir -p $OUT
export GRAFT_COMMIT=$(cat tools/commit_id.txt 2>/dev/null || echo worktree)
python tools/csrc_hash.py > $OUT/csrc_sha.txt
timeout -k 5 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
cp gpurun_out/r04_parity_at_scale.json $OUT/parity_at_scale.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
for PP in 1 0; do
  timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$PP -- python $ROOT/bench.py --steps 2 --warmup 0 --pools $PP --no-extras --no-cpu-baseline > $OUT/trace$PP.log 2>&1
  echo "trace pools=$PP rc=$?"
  for f in $(find $OUT/trace$PP -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_pools$PP.csv; done
  rm -rf $OUT/trace$PP
  grep '^{' $OUT/trace$PP.log | tail -1 > $OUT/bench_under_rocprof_pools$PP.json
done
cd $ROOT
bash tools/pmc_hbm.sh r04_$TAG 4096 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_r04_$TAG/summary.json $OUT/pmc_summary.json
# the bench lines LAST, with this session's PMC summary in place (profiles/r04_pmc_hbm_v9_session.json: same kernel-source hash -> not stale)
cp $OUT/pmc_summary.json profiles/r04_pmc_hbm_v9_session.json
timeout -k 5 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench default rc=$?"
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
timeout -k 5 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_driver.log 2>&1; echo "bench driver-style rc=$?"
grep '^{' $OUT/bench_driver.log | tail -1 > $OUT/bench_driver.json
rm -f profiles/r04_pmc_hbm_v9_session.json
# in-kernel phase timers of the same sources (-DIPM_PROFILE build made before the session: build/prof_final.so)
[ -f build/prof_final.so ] && timeout 300 python tools/phase_prof_scvx.py build/prof_final.so 4096 3 > $OUT/phase_prof.txt 2>&1
(cd scpp_amd/host && timeout 300 ./scvx_multi_gpu --batch 4096 --gpus 1 --slots 4096 --config ../config > $OUT/scvx_multi_gpu.log 2>&1; echo "scvx_multi_gpu rc=$?"; tail -3 $OUT/scvx_multi_gpu.log)
python - <<PY
import json
for n in ("bench_default","bench_driver","bench_under_rocprof_pools1","bench_under_rocprof_pools0"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); r=d["roofline"]
        print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "frac", round(r["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],3), "launches", r["launches"], "traffic", r["traffic"], "stale", (r.get("traffic_source") or {}).get("stale"))
    except Exception as e: print(n, "failed", e)
try:
    d=json.load(open("$OUT/bench_default.json")); c=d["config"]
    print(json.dumps({k:c.get(k) for k in ("single_pool","single_batch","step_length_rule","sc_mode","mpc_mode","parity")},indent=0)[:4000]); print(json.dumps(d.get("cpu_baseline"))[:1500])
    p=json.load(open("$OUT/pmc_summary.json")); print("PMC", p.get("ipm_bytes_per_instance_iteration"), p.get("calibration"), p.get("ipm_l2_hit_rate"), p.get("commit"), p.get("csrc_sha"))
except Exception as e: print("summary failed", e)
PY
head -8 $OUT/kernel_stats_pools1.csv | cut -c1-170
head -8 $OUT/kernel_stats_pools0.csv | cut -c1-170
