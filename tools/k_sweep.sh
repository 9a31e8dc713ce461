#!/bin/bash
# robustness / rate across horizons: the headline workload (SCvx, streaming engine) and SC mode at other K.   usage: k_sweep.sh [K ...]
for k in ${@:-15 30 64}; do
  timeout 400 python bench.py --K $k --batch 4096 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > /tmp/ks.json
  K=$k python - <<'PY'
import json, os
d = json.load(open('/tmp/ks.json')); c = d["config"]; s = c.get("sc_mode", {})
print("K", os.environ["K"], "converged SCvx traj/s %.0f" % d["value"], "converged fraction", c["converged_fraction"], "solver failures", c["solver_failures"],
      "ipm/traj %.0f" % c["mean_ipm_iterations_per_trajectory"], "| SC mode traj/s %.0f" % s.get("terminated_trajectories_per_s", 0), "failures", s.get("solver_failures"))
PY
done
