#!/bin/bash
# round-5 session 6: persistent engine at the bench configuration: step shares, slot counts, the driver-style job
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s6; mkdir -p $O
run() { local name=$1; shift; local envs=$1; shift
  env $envs timeout 600 python bench.py --no-extras --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); c=d["config"]; r=d["roofline"]; p=c["stream_profile_ticks"]; n=c["instances_timed"]
    tot=sum(p.values()) or 1
    print("$name", round(d["value"],1), "conv", c["converged_fraction"], "Mcycles/traj", {k: round(v/n/1e6,1) for k,v in p.items()}, "total", round(tot/n/1e6,1))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-600:])
PY
}
run e1_s8_b8192 SCPP_STREAM_ENGINE=1 --steps 8 --warmup 2
run e1_s8_b2048 SCPP_STREAM_ENGINE=1 --steps 32 --warmup 8 --batch 2048
run e1_s8_b4096 SCPP_STREAM_ENGINE=1 --steps 16 --warmup 4 --batch 4096
run e1_s4_default SCPP_STREAM_ENGINE=1
run e0_s4_default SCPP_STREAM_ENGINE=0
run e1_driver SCPP_STREAM_ENGINE=1 --steps 20 --warmup 5
run e0_driver SCPP_STREAM_ENGINE=0 --steps 20 --warmup 5
