#!/bin/bash
# round 6, session 14: primal and dual step lengths of their own (IPM_SPLIT_STEPS, the shipped library) against ECOS's common step length (build/common_step.so =
# the same sources with -DIPM_SPLIT_STEPS=0): GPU parity suite on the shipped library, then an alternating same-box A/B of the driver-style bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s14; mkdir -p $OUT
timeout -k 5 2700 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
cp gpurun_out/r06_parity_at_scale.json $OUT/parity_at_scale.json 2>/dev/null
bash tools/r06_ab.sh split_steps 3 --steps 6 --warmup 1 -- common_step shipped 2>&1 | tail -9
