#!/bin/bash
# usage: tools/ab.sh lib1.so lib2.so ...   -> one bench line (batch AB_BATCH, default 2048) per library
for lib in "$@"; do
  echo "== $lib"
  SCPP_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --batch ${AB_BATCH:-2048} --steps 1 --warmup 0 --no-cpu-baseline > /tmp/ab.log 2>&1
  grep '^{' /tmp/ab.log | python -c "
import json,sys
s=sys.stdin.read()
if not s.strip(): print('NO OUTPUT'); sys.exit()
d=json.loads(s)
print('traj/s %.1f  ipm ms/launch %.1f  disc ms/launch %.1f  roofline %.4f  fails %d' % (d['value'], d['roofline']['avg_launch_ms'], d['kernels']['discretize']['avg_launch_ms'], d['roofline']['frac'], d['config']['solver_failures']))" || tail -5 /tmp/ab.log
  grep -q '^{' /tmp/ab.log || tail -5 /tmp/ab.log
done
