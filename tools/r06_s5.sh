#!/bin/bash
# round 6, session 5: the persistent kernel for Rocket2D and zero-order hold on hardware (bitwise against the pool engine, throughput of both)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s5; mkdir -p $OUT
timeout -k 5 900 python tests/tools/engine_equal_variants.py 8192 2048 $OUT/engine_equal_variants.json > $OUT/engine_equal_variants.log 2>&1; echo "variants rc=$?"; cat $OUT/engine_equal_variants.log | tail -20
