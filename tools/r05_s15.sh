#!/bin/bash
# round-5 session 15: does the slot count (= allocation sizes / base addresses; the resident 2048 wavefronts are the same) move the persistent kernel?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s15; mkdir -p $O
for rep in 1 2; do
for cfg in "2048 32 8" "3072 21 5" "4096 16 4" "6144 11 3" "8192 8 2" "2560 26 6"; do
  set -- $cfg
  timeout 400 python bench.py --batch $1 --steps $2 --warmup $3 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json")); c=d["config"]; p=c["stream_profile_ticks"]; n=c["instances_timed"]
    print("slots $1 steps $2 rep $rep:", round(d["value"],1), "instances", n, {k: round(v/n/1e6,1) for k,v in p.items()})
except Exception as e: print("$1 failed", e, open("$O/b.err").read()[-300:])
PY
done
done
