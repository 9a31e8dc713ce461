#!/bin/bash
# round 6, session 11: update fused into the residual pass + dd copy kept on re-solves: bitwise vs the library before them, alternating A/B, then the driver's GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s11; mkdir -p $OUT
timeout -k 5 600 python tests/tools/lib_equal.py build/r06_base2.so build/r06_fused.so > $OUT/lib_equal.log 2>&1; echo "lib_equal rc=$?"; tail -3 $OUT/lib_equal.log
bash tools/r06_ab.sh s11 3 --steps 4 --warmup 1 -- r06_base2 r06_fused 2>&1 | tail -8
bash tools/r06_gpu_suite.sh
