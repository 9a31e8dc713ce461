#!/bin/bash
# rocprofv3 kernel statistics of the linear-MPC leg alone: 10 launches of mpc_solve_kernel at 32768 controllers + the
# 300-step closed loops of 4096 controllers (tests/tools/mpc_rate.py).
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_mpc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/tests/tools/mpc_rate.py > $OUT/trace.log 2>&1
echo "trace rc=$?"
tail -3 $OUT/trace.log
cd $ROOT
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-200; cp $f $OUT/kernel_stats.csv; done
