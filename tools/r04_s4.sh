#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04_s4; mkdir -p $O
timeout -k 5 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "at_scale" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^$" $O/pytest.log | tail -8 | cut -c1-700
cp gpurun_out/r04_parity_at_scale.json $O/parity_at_scale.json 2>/dev/null
REPS=3 bash tools/r04_ab.sh div1 r04c div1
