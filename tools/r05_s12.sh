#!/bin/bash
# round-5 session 12: the new / changed GPU tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s12; mkdir -p $O
timeout -k 5 2400 python -m pytest tests -m gpu -q -s -k "size_of_the_metric or sc_rocket2d_and_zero or emulator_and_gpu or persistent_engine_rows or bench_configuration or rocket2d_scvx_on_gpu or zero_order_hold_on_gpu" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -v "^$" $O/pytest_new.log | tail -25 | cut -c1-330
