#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile.sh [batch]
# rocprofv3 kernel-trace statistics + separate PMC passes (FETCH_SIZE / WRITE_SIZE) of one bench step.
# Outputs under gpurun_out/prof/ (copy the summaries into profiles/ afterwards).
B=${1:-2048}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
echo "pmc write rc=$?"
cd $ROOT
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("gpurun_out/prof/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", "?"))
            acc[k][0] += float(r.get("Counter_Value", 0)); acc[k][1] += 1
        for k, v in sorted(acc.items()):
            print(tag, k, "sum=%.4g" % v[0], "n=%d" % v[1], "per_dispatch=%.4g" % (v[0] / max(v[1], 1)))
PY
