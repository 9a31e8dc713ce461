#!/bin/bash
# round-4 session 2: the whole GPU suite on the restructured ipm_kernel + smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04_s2; mkdir -p $O
timeout -k 5 2400 python -m pytest tests -m gpu -q -s -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
