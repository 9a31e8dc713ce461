#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s10; mkdir -p $O
timeout 300 python tests/tools/r2d_audit_rows.py > $O/r2d_rows.log 2>&1; echo "rows rc=$?"; cat $O/r2d_rows.log | cut -c1-260
timeout -k 5 2400 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_parity.py::test_rocket2d_scvx_on_gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log | cut -c1-400
