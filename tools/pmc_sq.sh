#!/bin/bash
# SQ issue/stall breakdown of the kernels of one small bench step (own PMC pass, no tracing)
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $OUT -- python $ROOT/bench.py --batch ${1:-2048} --steps 1 --warmup 0 --no-cpu-baseline --scvx-batch 0 --mpc-batch 0 > $OUT/log.txt 2>&1
echo rc=$?
cd $ROOT
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, d in acc.items():
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        print(k)
        for c, v in sorted(d.items()):
            print("   %-22s %.4g  (%.1f%% of wave cycles)" % (c, v, 100 * v / wc))
PY
