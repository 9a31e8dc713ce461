#!/bin/bash
# The kernel sources compiled for the CPU wave emulator with AddressSanitizer + UBSan, and every emulator-backed test run on that
# build (about 8 min to compile, 5 min to run).  Round 3: found one read past a node's NX states in discretize_kernel (lanes
# row >= NX; past the end of X at the last node of a Rocket2D batch), fixed; clean since.     usage: bash tools/asan_emu.sh
cd "$(dirname "$0")/.." || exit 1
OUT=${TMPDIR:-/tmp}/scpp_asan; mkdir -p $OUT
g++ -O1 -g -std=c++17 -fPIC -DSCPP_HIP_EMU -fsanitize=address,undefined -fno-omit-frame-pointer -Itests/emu -shared \
    -o $OUT/libscpp_emu_asan.so -x c++ scpp_amd/csrc/scpp_hip.cpp 2> $OUT/build.log || { tail $OUT/build.log; exit 1; }
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" SCPP_EMU_LIBRARY=$OUT/libscpp_emu_asan.so \
    python -m pytest tests/test_emu_kernels.py tests/test_emu_split.py tests/test_emu_stream_fuzz.py tests/test_emu_batch_independence.py tests/test_mpc.py tests/test_abi_errors.py tests/test_sc_loop_pin.py tests/test_subproblem_pin.py \
    -q -m "not gpu" > $OUT/run.log 2>&1
echo "pytest rc=$?"; grep -c "ERROR: AddressSanitizer\|runtime error" $OUT/run.log; tail -2 $OUT/run.log
# the oracle (test infrastructure) under the same sanitizers
g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $OUT/liboracle_asan.so oracle/capi.cpp 2>> $OUT/build.log
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" SCPP_ORACLE_LIBRARY=$OUT/liboracle_asan.so \
    python -m pytest tests/test_oracle_discretization.py tests/test_oracle_scvx.py tests/test_oracle_socp.py tests/test_oracle_sc.py tests/test_oracle_model.py \
    tests/test_oracle_mpc.py tests/test_oracle_rkf78.py tests/test_oracle_sc_sim.py -q -m "not gpu" > $OUT/run_oracle.log 2>&1
echo "oracle pytest rc=$?"; grep -c "ERROR: AddressSanitizer\|runtime error" $OUT/run_oracle.log; tail -2 $OUT/run_oracle.log
