#!/bin/bash
# round-5 session 4: hardware queues vs slot pools (HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues, default 4: six pool streams share four queues)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s4; mkdir -p $O
run() { # name, env..., -- args
  local name=$1; shift
  env "$@" timeout 400 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); c=d["config"]; r=d["roofline"]
    print("$name", round(d["value"],1), "ipm span", round(r["avg_launch_ms"],2), "launches", r["launches"], "disc ms", round(d["kernels"]["discretize"]["avg_launch_ms"],2), "pools", c.get("pools"))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-400:])
PY
}
EXTRA="" run q4_p6_a A=1
EXTRA="" run q8_p6_a GPU_MAX_HW_QUEUES=8
EXTRA="--pools 8" run q8_p8_a GPU_MAX_HW_QUEUES=8
EXTRA="--pools 4" run q4_p4_a A=1
EXTRA="" run q4_p6_b A=1
EXTRA="" run q8_p6_b GPU_MAX_HW_QUEUES=8
EXTRA="--pools 8" run q8_p8_b GPU_MAX_HW_QUEUES=8
EXTRA="--pools 7" run q8_p7_b GPU_MAX_HW_QUEUES=8
