#!/bin/bash
# the driver's GPU commands on the current tree: pytest -m gpu, smoke()
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_suite; mkdir -p $OUT
timeout -k 5 2700 python -m pytest tests -x -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log | cut -c1-300
