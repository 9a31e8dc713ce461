#!/bin/bash
# round 6, session 13: placement selection (SCvxAlgorithm.initialize(placement_candidates)): bench.py with 1 and with 4 candidates, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s13; mkdir -p $OUT
for r in 1 2 3 4; do for n in 1 4; do
  timeout -k 5 600 python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --placement-candidates $n > $OUT/c$n.$r.log 2>&1
  grep '^{' $OUT/c$n.$r.log | tail -1 > $OUT/c$n.$r.json
  python - $OUT/c$n.$r.json $n $r <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d["config"]["placement"]
print("candidates %s round %s value %.1f" % (sys.argv[2], sys.argv[3], d["value"]), "probes", [round(max(r[1:])) for r in p.get("trajectories_per_s_of_the_probes", [])], "chosen", p.get("chosen"))
PY
done; done
