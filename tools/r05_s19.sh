#!/bin/bash
# round-5 session 19 (experiment): do the two speed regimes follow the workspace's virtual-address alignment?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s19; mkdir -p $O
run() { local name=$1; shift; local envs=$1; shift
  env $envs timeout 400 python bench.py --no-extras --no-cpu-baseline --library $PWD/build/dbg_alloc.so "$@" > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); c=d["config"]; p=c["stream_profile_ticks"]; n=c["instances_timed"]
    print("$name:", round(d["value"],1), "solve Mcycles/traj", round(p["solve"]/n/1e6,1), [l.strip() for l in open("$O/b.err") if l.startswith("ws")][:1])
except Exception as e: print("$name failed", e, open("$O/b.err").read()[-300:])
PY
}
for rep in 1 2; do
run "8192 default   " SCPP_DEBUG_ALLOC=1 --steps 8 --warmup 2
run "8192 align 1GB " SCPP_WS_ALIGN_MB=1024 --steps 8 --warmup 2
run "8192 align 64MB" SCPP_WS_ALIGN_MB=64 --steps 8 --warmup 2
run "4096 default   " SCPP_DEBUG_ALLOC=1 --batch 4096 --steps 16 --warmup 4
run "4096 align 1GB " SCPP_WS_ALIGN_MB=1024 --batch 4096 --steps 16 --warmup 4
run "6144 default   " SCPP_DEBUG_ALLOC=1 --batch 6144 --steps 11 --warmup 3
done
