#!/bin/bash
# instruction-cache behaviour of the persistent SCvx kernel (own PMC passes, no tracing): bash tools/r06_icache.sh [batch] [steps]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_icache; mkdir -p $OUT; B=${1:-8192}; ST=${2:-2}
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout -k 5 400 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python $ROOT/bench.py --batch $B --steps $ST --warmup 0 --no-cpu-baseline --no-extras > $OUT/$n.log 2>&1; echo "$n rc=$?"; }
run p1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run p2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
run p3 SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM
cd $ROOT
python - <<'PY' | tee gpurun_out/r06_icache/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/r06_icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "persistent" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-30s %.5g" % (c, v))
    if d.get("SQC_ICACHE_REQ"): print("   icache hit rate %.4f, misses per VALU instruction %.5f" % (d.get("SQC_ICACHE_HITS", 0) / d["SQC_ICACHE_REQ"], d.get("SQC_ICACHE_MISSES", 0) / max(d.get("SQ_INSTS_VALU", 0), 1)))
    if d.get("SQ_WAVE_CYCLES"): print("   of wave cycles: waiting for instruction issue %.3f, issuing %.3f, waiting any %.3f" % (d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"], d.get("SQ_ACTIVE_INST_ANY", 0) / d["SQ_WAVE_CYCLES"], d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]))
    if d.get("SQ_IFETCH_LEVEL") and d.get("SQ_IFETCH"): print("   average instruction-fetch latency (level / fetches): %.1f quad-cycles?" % (d["SQ_IFETCH_LEVEL"] / d["SQ_IFETCH"]))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
