#!/bin/bash
# closing GPU session of round 4: (1) SC mode A/B + bitwise (previous commit's build vs this one), (2) headline A/B with this build FIRST in each
# pair (the earlier A/B had it second), (3) the measurement session of tools/r04_final.sh for the shipped library.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_s5
timeout 200 python tests/tools/sc_mode_ab.py build/sc_head.so build/sc_v3.so 8192 2 > gpurun_out/r04_s5/sc_mode_ab.log 2>&1; cat gpurun_out/r04_s5/sc_mode_ab.log
REPS=2 STEPS=8 bash tools/r04_ab.sh headline_v3 sc_v3 sc_head | tee gpurun_out/r04_s5/headline_ab.log
bash tools/r04_final.sh final5
