#!/bin/bash
# round-5 session 9: is the Rocket2D SCvx audit failure a regression?  the round-4 tree against HEAD on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; O=$ROOT/gpurun_out/r05_s9; mkdir -p $O
(cd build/r04tree && timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_rocket2d_scvx_on_gpu" > $O/r04_r2d.log 2>&1; echo "r04 tree rc=$?"; tail -4 $O/r04_r2d.log)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_rocket2d_scvx_on_gpu" > $O/head_r2d.log 2>&1; echo "head rc=$?"; tail -4 $O/head_r2d.log | cut -c1-300
(cd build/r04tree && timeout 300 python $ROOT/tests/tools/dump_runs.py $O/r04.npz 2>&1 | tail -1)
timeout 300 python tests/tools/dump_runs.py $O/head.npz 2>&1 | tail -1
python - <<PY
import numpy as np
a=np.load("$O/r04.npz"); b=np.load("$O/head.npz")
for k in a.files:
    eq = np.array_equal(a[k], b[k])
    print(k, "bitwise equal" if eq else "DIFFERS max|d|=%g" % (np.abs(a[k].astype(float)-b[k].astype(float)).max()))
PY
