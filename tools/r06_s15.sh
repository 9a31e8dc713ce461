#!/bin/bash
# round 6, session 15 (split step lengths): the closed-loop test with its restated planned-time check, the 4096-instance parity audit of the headline mode, the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s15; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "stop_rule or lander" > $OUT/pytest_closed_loop.log 2>&1; echo "closed loop rc=$?"; grep -E "SC_sim closed|passed|failed" $OUT/pytest_closed_loop.log | cut -c1-400
SCPP_PARITY_N=4096 SCPP_PARITY_FIRST=400000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "at_scale_parity" > $OUT/pytest_parity4096.log 2>&1; echo "parity4096 rc=$?"; grep -E "SCvx at scale|passed|failed" $OUT/pytest_parity4096.log | cut -c1-600
cp gpurun_out/r06_parity_at_scale_N4096_first400000.json $OUT/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_full.log 2>&1; echo "bench rc=$?"; grep '^{' $OUT/bench_driver_full.log | tail -1 > $OUT/bench_driver_full.json
python -c "
import json; d=json.load(open('$OUT/bench_driver_full.json')); print('value', d['value'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])"
