#!/bin/bash
# copies the files of a closing session (tools/r06_final.sh <tag>, merged back into gpurun_out/r06_<tag>/) into profiles/ under the round's names
#   usage: bash tools/r06_import_final.sh <tag> <pmc version number>      e.g.  final2 2  ->  profiles/r06_pmc_hbm_v2_final.json
cd "$(dirname "$0")/.." || exit 1
TAG=${1:?tag}; V=${2:?pmc version}; S=gpurun_out/r06_$TAG; P=profiles
[ -f $S/pmc_summary.json ] || { echo "no $S/pmc_summary.json"; exit 1; }
cp $S/bench_default.json $P/r06_bench_${TAG}_default.json
cp $S/bench_driver.json $P/r06_bench_${TAG}_driver.json
cp $S/bench_driver_pools.json $P/r06_bench_${TAG}_driver_pool_engine.json
cp $S/bench_under_rocprof_persistent.json $P/r06_bench_${TAG}_under_rocprof_persistent.json
cp $S/bench_under_rocprof_pools1.json $P/r06_bench_${TAG}_under_rocprof_single_pool.json
cp $S/kernel_stats_persistent.csv $P/r06_kernel_stats_${TAG}_persistent.csv
cp $S/kernel_stats_pools1.csv $P/r06_kernel_stats_${TAG}_single_pool.csv
cp $S/pmc_summary.json $P/r06_pmc_hbm_v${V}_final.json
cp $S/parity_at_scale.json $P/r06_parity_at_scale.json
cp $S/pytest_gpu.log $P/r06_pytest_gpu_${TAG}.log
cp $S/smoke.log $P/r06_smoke_${TAG}.log
cp $S/scvx_multi_gpu.log $P/r06_scvx_multi_gpu_one_rank.log
python - <<PY
import json
p=json.load(open("$P/r06_pmc_hbm_v${V}_final.json")); q=json.load(open("$P/r06_parity_at_scale.json"))
print("pmc csrc_sha", p.get("csrc_sha"), "commit", p.get("commit"), "| parity csrc_sha", q.get("csrc_sha"), "| here", open("$S/csrc_sha.txt").read().strip())
PY
