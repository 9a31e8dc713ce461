#!/bin/bash
# round-5 session 21: the persistent engine at other horizons (K = 15 shipped SC.info, 30 shipped SCvx.info, 50 BASELINE, 64 the layout's limit), both engines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for k in 15 30 50 64; do
for E in 1 0; do
  SCPP_STREAM_ENGINE=$E timeout 400 python bench.py --K $k --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > /tmp/ks.json
  K=$k E=$E python - <<'PY'
import json, os
d = json.load(open('/tmp/ks.json')); c = d["config"]
print("K", os.environ["K"], "engine", "persistent" if os.environ["E"] == "1" else "pools     ", "converged SCvx traj/s %.0f" % d["value"], "converged fraction %.4f" % c["converged_fraction"], "solver failures", c["solver_failures"],
      "ipm/traj %.0f" % c["mean_ipm_iterations_per_trajectory"], "solves %.1f" % c["mean_subproblem_solves"])
PY
done
done
