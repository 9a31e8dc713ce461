#!/bin/bash
# same-box A/B of two builds of the library: alternating short bench runs.   usage: r02_pair.sh <base.so> <new.so> [rounds]
A=$PWD/$1; B=$PWD/$2; R=${3:-3}
for r in $(seq 1 $R); do
  for L in $A $B; do
    python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --library $L 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], 'value', round(d['value'],1), 'ipm avg ms', round(d['roofline']['avg_launch_ms'],2), 'disc avg ms', round(d['kernels']['discretize']['avg_launch_ms'],2))"
  done
done
