#!/bin/bash
# same-box alternating A/B of alternative libraries (build/<name>.so, tools/r06_variant.sh): tools/r06_ab.sh <tag> <rounds> <bench args...> -- name1 name2 ...
#   -> gpurun_out/r06_ab_<tag>.json ; one bench line per (round, library), driver-style flags by default
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=$1; ROUNDS=$2; shift 2
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
[ ${#ARGS[@]} -eq 0 ] && ARGS=(--steps 4 --warmup 1)
OUT=gpurun_out/r06_ab_$TAG; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for n in "$@"; do
    LIB=$PWD/build/$n.so; [ "$n" = shipped ] && LIB=$PWD/scpp_amd/libscpp_hip.so
    SCPP_HIP_LIBRARY=$LIB timeout -k 5 600 python bench.py "${ARGS[@]}" --no-extras --no-cpu-baseline --library $LIB > $OUT/$n.$r.log 2>&1
    grep '^{' $OUT/$n.$r.log | tail -1 > $OUT/$n.$r.json
    python - "$OUT/$n.$r.json" "$n" "$r" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; s = (d["roofline"].get("steps") or {})
    print("%-16s round %s  value %8.1f  ipm/traj %.2f solves %.3f conv %.4f fails %d  solve Mcyc/traj %.0f" % (sys.argv[2], sys.argv[3], d["value"], c["mean_ipm_iterations_per_trajectory"],
          c["mean_subproblem_solves"], c["converged_fraction"], c["solver_failures"], (s.get("solve") or {}).get("Mcycles_per_trajectory", 0)))
except Exception as e:
    print(sys.argv[2], "round", sys.argv[3], "FAILED", e)
PY
  done
done
python - "$OUT" "$TAG" "$@" <<'PY'
import json, sys, glob, os
out, tag, names = sys.argv[1], sys.argv[2], sys.argv[3:]
S = {"tag": tag, "libraries": {}}
for n in names:
    vals = []
    for f in sorted(glob.glob(f"{out}/{n}.*.json")):
        try: vals.append(json.load(open(f))["value"])
        except Exception: pass
    S["libraries"][n] = {"converged_per_s": vals, "mean": sum(vals) / len(vals) if vals else None}
json.dump(S, open(f"gpurun_out/r06_ab_{tag}.json", "w"), indent=1)
print(json.dumps(S))
PY
