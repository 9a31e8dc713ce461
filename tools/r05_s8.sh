#!/bin/bash
# round-5 session 8: the whole GPU suite on the refactored kernels + persistent default, then the full default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s8; mkdir -p $O
timeout -k 5 2400 python -m pytest tests -m gpu -q -s -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout -k 5 900 python bench.py > $O/bench_default.log 2>&1; echo "bench default rc=$?"
grep '^{' $O/bench_default.log | tail -1 > $O/bench_default.json
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); c=d["config"]; r=d["roofline"]
print("value", round(d["value"],1), "frac", round(r["frac"],4), "kernel", r["kernel"][:40], "launches", r["launches"])
print(json.dumps(r.get("steps"), indent=0)[:1500])
print(json.dumps(c.get("what_converged_means"), indent=0)[:1800])
print({k: (c.get(k) or {}).get("converged_trajectories_per_s") for k in ("single_pool","single_batch","step_length_rule")})
print(json.dumps(d.get("cpu_baseline"))[:600])
PY
