#!/bin/bash
# round 6, session 1: where HEAD stands on this round's boxes -- bench line, in-kernel phase profile (SCvx mode), PMC summary at the bench's own shape
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_s1; mkdir -p $OUT
export GRAFT_COMMIT=$(cat tools/commit_id.txt 2>/dev/null || echo worktree)
timeout -k 5 600 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $OUT/bench_head.log 2>&1; echo "bench rc=$?"
grep '^{' $OUT/bench_head.log | tail -1 > $OUT/bench_head.json
python - <<PY
import json
d=json.load(open("$OUT/bench_head.json")); r=d["roofline"]
print("HEAD value", round(d["value"],1), "frac", round(r["frac"],4), "kernel_s", r["kernel_time_s"], "ipm/traj", d["config"]["mean_ipm_iterations_per_trajectory"], "fails", d["config"]["solver_failures"])
print(json.dumps(r.get("steps"))[:1200])
PY
timeout -k 5 600 python tools/phase_prof_scvx.py build/r06_prof.so 4096 6 > $OUT/phase_prof_scvx.txt 2>&1; echo "phase prof rc=$?"; cat $OUT/phase_prof_scvx.txt | tail -20
SCPP_STREAM_ENGINE=0 timeout -k 5 600 python tools/phase_prof_scvx.py build/r06_prof.so 4096 6 > $OUT/phase_prof_scvx_pools.txt 2>&1; echo "phase prof (pool engine) rc=$?"; cat $OUT/phase_prof_scvx_pools.txt | tail -20
bash tools/pmc_hbm.sh r06_s1 8192 2 > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_r06_s1/summary.json $OUT/pmc_summary.json
python - <<PY
import json
p=json.load(open("$OUT/pmc_summary.json"))
print({k:p.get(k) for k in ("engine","ipm_bytes_per_instance_iteration","bytes_per_trajectory","calibration","ipm_l2_hit_rate","csrc_sha","trajectories","ipm_iterations")})
print(p.get("ipm_mfma")); print(p.get("sq")); print(p.get("tcc"))
PY
