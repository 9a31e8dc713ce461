#!/bin/bash
# round 6, session 12: the fused update with the directions arriving in pieces (scratch 444 -> 172 B per lane): bitwise, alternating A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s12; mkdir -p $OUT
timeout -k 5 600 python tests/tools/lib_equal.py build/r06_base2.so build/r06_fused2.so > $OUT/lib_equal.log 2>&1; echo "lib_equal rc=$?"; tail -3 $OUT/lib_equal.log
bash tools/r06_ab.sh s12 3 --steps 4 --warmup 1 -- r06_base2 r06_fused2 2>&1 | tail -8
