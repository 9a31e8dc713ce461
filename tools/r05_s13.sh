#!/bin/bash
# round-5 session 13: -ffp-contract=on (fused multiply-adds formed per source expression, not by the optimiser's context): are the two engines bitwise
# equal then, and what does it cost?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s13; mkdir -p $O
SCPP_HIP_LIBRARY=$PWD/build/contract_on.so timeout 300 python tests/tools/engine_equal.py 3000 1024 > $O/engine_equal_on.log 2>&1; echo "engine_equal (contract=on) rc=$?"; tail -5 $O/engine_equal_on.log | cut -c1-250
for rep in 1 2; do
for L in scpp_amd/libscpp_hip.so build/contract_on.so; do
  timeout 400 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --library $PWD/$L > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); c=d["config"]; p=c["stream_profile_ticks"]; n=c["instances_timed"]
    print("$L rep $rep", round(d["value"],1), "conv", c["converged_fraction"], "ipm/traj", round(c["mean_ipm_iterations_per_trajectory"],2), {k: round(v/n/1e6,1) for k,v in p.items()})
except Exception as e: print("$L failed", e, open("$O/b.err").read()[-500:])
PY
done
done
