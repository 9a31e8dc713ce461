#!/bin/bash
# round-5 session 3: kernel trace of the split schedule (single pool and default pools): per-dispatch durations and gaps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; O=$ROOT/gpurun_out/r05_s3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for PP in 1 0; do
  SCPP_IPM_SCHEDULE=1 timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$PP -- python $ROOT/bench.py --steps 1 --warmup 0 --pools $PP --no-extras --no-cpu-baseline > $O/trace$PP.log 2>&1
  echo "trace pools=$PP rc=$?"
  for f in $(find $O/trace$PP -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_pools$PP.csv; done
  for f in $(find $O/trace$PP -name "*kernel_trace.csv"); do python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
print(len(rows), "dispatches", list(rows[0].keys()))
# compact: kernel short name, start, end, stream/queue
import re
out = open("$O/dispatches_pools$PP.csv", "w")
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    n = r["Kernel_Name"]
    s = "A" if "ipm_split_kernel" in n else "F" if "ipm_factor_kernel" in n else "D" if "discretize" in n else "C" if "cost_update" in n else "R" if "refill" in n else "o"
    out.write(f"{s},{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - t0},{r.get('Queue_Id','')},{r.get('Stream_Id','')}\n")
out.close()
PY
  done
  rm -rf $O/trace$PP
  grep '^{' $O/trace$PP.log | tail -1 > $O/bench_under_rocprof_pools$PP.json
  head -6 $O/kernel_stats_pools$PP.csv | cut -c1-200
done
