#!/bin/bash
# where the wave cycles of ipm_kernel go: issue activity by instruction class, MFMA pipe, LDS / instruction-issue waits (own PMC
# passes, no tracing).   usage: bash tools/pmc_sq2.sh [batch]
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_sq2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/avail_sq.txt
B=${1:-4096}
run() { n=$1; shift
  timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python $ROOT/bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline --no-extras --pools 1 > $OUT/$n.log 2>&1; echo "$n rc=$?"; }
run a SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_EXP_GDS
run b SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
run c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU
cd $ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_sq2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] = max(acc[r["Kernel_Name"][:40]][r["Counter_Name"]], 0) + float(r["Counter_Value"])
for k, d in acc.items():
    if "ipm_kernel" not in k: continue
    wc = d.get("SQ_WAVE_CYCLES", 1) / 3.0  # collected in three passes
    print(k, "wave cycles per pass %.4g" % wc)
    for c, v in sorted(d.items()):
        if c != "SQ_WAVE_CYCLES": print("   %-32s %.5g   (%.3f of wave cycles)" % (c, v, v / wc))
PY
