#!/bin/bash
# round 6, session 16: the repeat of a failed split-step attempt with the common step length (soak instance 454733) and the linear MPC kernel's split step lengths
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s16; mkdir -p $OUT
timeout 600 python tools/r06_trace_failure.py > $OUT/trace_454733.log 2>&1; tail -4 $OUT/trace_454733.log | cut -c1-200
timeout -k 5 2700 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 900 python tools/r06_mpc_ab.py > $OUT/mpc_ab.log 2>&1; echo "mpc ab rc=$?"; tail -7 $OUT/mpc_ab.log | cut -c1-250
