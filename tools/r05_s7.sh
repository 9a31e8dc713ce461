#!/bin/bash
# round-5 session 7: persistent engine with the LDS-resident segment fields (dynamic LDS shared with the integration)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s7; mkdir -p $O
timeout 300 python tests/tools/engine_equal.py 3000 1024 > $O/engine_equal.log 2>&1; echo "engine_equal rc=$?"; tail -4 $O/engine_equal.log
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
run e1_a SCPP_STREAM_ENGINE=1 --steps 8 --warmup 2
run e0_a SCPP_STREAM_ENGINE=0 --steps 8 --warmup 2
run e1_b SCPP_STREAM_ENGINE=1 --steps 8 --warmup 2
run e0_b SCPP_STREAM_ENGINE=0 --steps 8 --warmup 2
