#!/bin/bash
# robustness / rate across horizons (SC mode + a small SCvx sub-run)
for k in ${@:-15 30 64}; do
  timeout 300 python bench.py --K $k --batch 4096 --steps 1 --warmup 0 --no-cpu-baseline --scvx-batch 1024 2>/dev/null | tail -1 > /tmp/ks.json
  K=$k python - <<'PY'
import json, os
d = json.load(open('/tmp/ks.json')); v = d["config"]["scvx_mode"]
print("K", os.environ["K"], "traj/s %.0f" % d["value"], "fails", d["config"]["solver_failures"], "ipm/traj %.0f" % d["config"]["mean_ipm_iterations_per_trajectory"],
      "| scvx converged", v.get("converged_fraction"), "fails", v.get("solver_failures"))
PY
done
