#!/bin/bash
# round 6: one million trajectories through the shipped library in ONE streaming job (128 steps x 8192): solver failures, convergence, rate over three minutes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_soak; mkdir -p $OUT
timeout -k 5 1500 python bench.py --steps 128 --warmup 1 --no-extras --no-cpu-baseline > $OUT/bench_1M.log 2>&1; echo "rc=$?"
grep '^{' $OUT/bench_1M.log | tail -1 > $OUT/bench_1M.json
python - <<PY
import json
d=json.load(open("$OUT/bench_1M.json")); c=d["config"]
print("value", d["value"], "instances", c["instances_timed"], "converged_fraction", c["converged_fraction"], "solver_failures", c["solver_failures"], "ms/step", d["ms_per_step"], "mean solves", c["mean_subproblem_solves"], "ipm/traj", c["mean_ipm_iterations_per_trajectory"])
PY
