#!/bin/bash
# instruction-cache behaviour of ipm_kernel (own PMC pass, no tracing): the code one interior-point iteration walks through is
# ~250 KB against a 64 KB instruction cache shared by two CUs.   usage: bash tools/pmc_icache.sh [batch]
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\b\|SQ_INSTS_SALU\|SQ_ACTIVE_INST_ANY\|SQ_WAIT_ANY" | sort -u > $OUT/avail.txt
echo "available:"; tr '\n' ' ' < $OUT/avail.txt; echo
run() { # name, counters...
  n=$1; shift
  timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python $ROOT/bench.py --batch ${B:-4096} --steps 1 --warmup 0 --no-cpu-baseline --no-extras --pools 1 > $OUT/$n.log 2>&1
  echo "$n rc=$?"
}
B=${1:-4096}
run p1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run p2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
cd $ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "ipm_kernel" not in k and "discretize" not in k and "cost_update" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-30s %.5g" % (c, v))
    if d.get("SQC_ICACHE_REQ"): print("   icache hit rate %.3f, misses per VALU instruction %.4f" % (d.get("SQC_ICACHE_HITS", 0) / d["SQC_ICACHE_REQ"], d.get("SQC_ICACHE_MISSES", 0) / max(d.get("SQ_INSTS_VALU", 0), 1)))
    if d.get("SQ_WAVE_CYCLES"): print("   of wave cycles: waiting for instruction %.3f, issuing %.3f, waiting any %.3f" % (d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"], d.get("SQ_ACTIVE_INST_ANY", 0) / d["SQ_WAVE_CYCLES"], d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]))
PY
