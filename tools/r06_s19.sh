#!/bin/bash
# round 6, session 19: warm-started SC solves on the common step length (SC_sim's shape, A/B against build/common_step.so), the closing session final7, the soak job
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r06_s19
timeout 900 python tools/r06_scsim_ab.py > gpurun_out/r06_s19/scsim_ab.log 2>&1; tail -4 gpurun_out/r06_s19/scsim_ab.log | cut -c1-330
cp gpurun_out/r06_ab_scsim_split_steps.json gpurun_out/r06_s19/scsim_ab_after.json
bash tools/r06_final.sh final7
bash tools/r06_soak.sh
