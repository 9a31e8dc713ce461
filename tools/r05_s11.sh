#!/bin/bash
# round-5 session 11: PMC bytes / counters and rocprofv3 kernel statistics of the persistent engine (mid-round)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; O=$ROOT/gpurun_out/r05_s11; mkdir -p $O
bash tools/pmc_hbm.sh r05_mid 4096 > $O/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_r05_mid/summary.json $O/pmc_summary.json
python - <<PY
import json
p=json.load(open("$O/pmc_summary.json"))
print({k:p.get(k) for k in ("engine","ipm_bytes_per_instance_iteration","bytes_per_trajectory","calibration","ipm_l2_hit_rate","csrc_sha")})
print(p.get("kernels")); print(p.get("ipm_mfma"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $ROOT/bench.py --steps 2 --warmup 0 --no-extras --no-cpu-baseline > $O/trace.log 2>&1; echo "trace rc=$?"
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
rm -rf $O/trace; grep '^{' $O/trace.log | tail -1 > $O/bench_under_rocprof.json
head -6 $O/kernel_stats.csv | cut -c1-220
