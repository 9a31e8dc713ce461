#!/bin/bash
# SQ issue/stall breakdown of mpc_solve_kernel (own PMC pass, no tracing): tests/tools/mpc_rate.py
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_mpc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -- python $ROOT/tests/tools/mpc_rate.py > $OUT/log.txt 2>&1
echo rc=$?
cd $ROOT
python - <<'PY' | tee gpurun_out/pmc_mpc/summary.txt
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_mpc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        print(k, "(dispatches: %d)" % max(n[(k, c)] for c in d))
        for c, v in sorted(d.items()):
            print("   %-22s %.4g  (%.1f%% of wave cycles)" % (c, v, 100 * v / wc))
PY
