#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04_s3; mkdir -p $O
timeout -k 5 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "zero_order_hold_on_gpu" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^$" $O/pytest.log | tail -25
