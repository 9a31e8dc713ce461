#!/bin/bash
# round-5 session 1: the prepared ablation (ipm_kernel without its factor sweep at 2 and 3 waves/SIMD: in-kernel phase timers) + a same-box
# reference bench of the shipped library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/r05_ablation_run.sh
mkdir -p gpurun_out/r05_s1
timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05_s1/bench_ref.json 2> gpurun_out/r05_s1/bench_ref.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/r05_s1/bench_ref.json")); c=d["config"]; r=d["roofline"]
print("ref", round(d["value"],1), "frac", round(r["frac"],4), "ipm span", round(r["avg_launch_ms"],2), "disc ms", round(d["kernels"]["discretize"]["avg_launch_ms"],2))
PY
