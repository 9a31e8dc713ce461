#!/bin/bash
# round 6, session 3: throughput against resident wavefronts per CU (the persistent kernel, one library), and the available memory-side counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s3; mkdir -p $OUT
for PAD in 0 3072 7168 12288 20480; do
  SCPP_PERSIST_LDS_PAD=$PAD timeout -k 5 600 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $OUT/pad_$PAD.log 2>&1
  grep '^{' $OUT/pad_$PAD.log | tail -1 > $OUT/pad_$PAD.json
  python - $OUT/pad_$PAD.json $PAD <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); s = d["roofline"]["steps"]
print("lds pad %6s  value %8.1f  solve %.0f disc %.0f cost %.0f Mcycles/traj" % (sys.argv[2], d["value"], s["solve"]["Mcycles_per_trajectory"], s["discretize"]["Mcycles_per_trajectory"], s["cost"]["Mcycles_per_trajectory"]))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mall|umc|hbm|dram|EA0|EA_|MC_|GCEA|RDREQ|WRREQ" | cut -c1-200 | head -80 > $OLDPWD/$OUT/counters.txt
cd $OLDPWD; wc -l $OUT/counters.txt; head -60 $OUT/counters.txt
