#!/bin/bash
# round-4 session 1: the new multi-pool GPU tests + a baseline bench line of the round's starting library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04_s1; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "bench_configuration or rocket2d_stream_multi_pool or adaptive_step or stream_equals_batch or default_pools" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 5 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"
grep '^{' $O/bench.log | tail -1 > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json")); r=d["roofline"]
print("value", round(d["value"],1), "frac", round(r["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],3), "stale", r["traffic_source"])
PY
