#!/bin/bash
# round 6, session 8: recorded iterates on hardware, SC_sim at size with the vectorised host loop, host executables, path audits (they read the record now)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/r06_s8; mkdir -p $OUT
timeout -k 5 1500 python -m pytest tests -m gpu -x -q -s -k "recorded_iterates or sc_sim_monte_carlo or host_executables or independent_path or persistent or broken_iterate" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|SC_sim at size|getAllSolutions|error" $OUT/pytest.log | tail -12
