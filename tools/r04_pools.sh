#!/bin/bash
# pool-count sweep of the streaming engine with the shipped library (same box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04_pools; mkdir -p $O
for rep in 1 2; do for P in "$@"; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --pools $P > $O/p${P}_$rep.json 2> $O/p${P}_$rep.err
  python -c "
import json
d=json.load(open('$O/p${P}_$rep.json')); print('pools $P rep $rep', round(d['value'],1), 'ipm span', round(d['roofline']['avg_launch_ms'],2), 'rounds', d['config']['rounds'])"
done; done
