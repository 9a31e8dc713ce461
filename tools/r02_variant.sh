#!/bin/bash
# bench + parity subset of an alternative build of the library.   usage: r02_variant.sh <lib.so> [more bench args]
LIB=$PWD/$1; shift
export SCPP_HIP_LIBRARY=$LIB
timeout -k 5 300 python -m pytest tests -m gpu -x -q -k "twin or scvx_batch or stream or rocket2d" 2>&1 | tail -2
timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --library $LIB "$@" 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$LIB', 'value', round(d['value'],1), 'ipm avg ms', round(d['roofline']['avg_launch_ms'],2), 'conv', d['config']['converged_fraction'], 'fail', d['config']['solver_failures'])"
