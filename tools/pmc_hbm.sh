#!/bin/bash
# HBM traffic, matrix-core and stall counters of the shipped kernels (rocprofv3 PMC; every counter group in its own pass, no
# tracing flags), with the FETCH_SIZE / WRITE_SIZE calibration of tools/hbm_calib.hip measured in the same session.
#   usage: bash tools/pmc_hbm.sh [tag] [batch] [steps]   -> gpurun_out/pmc_<tag>/summary.json (copy to profiles/rNN_pmc_hbm_v<n>_<tag>.json)
#   (round 6: `steps` batches as one job -- 8192 2 is the bench's own shape: 8192 slots, the queue refilled once)
TAG=${1:-v1}; BATCH=${2:-4096}; STEPS=${3:-1}
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
# the calibration kernels (tools/bin/ is not tracked: built here if the checkout is fresh)
[ -x $ROOT/tools/bin/hbm_calib ] || { mkdir -p $ROOT/tools/bin && hipcc --offload-arch=gfx950 -O3 -o $ROOT/tools/bin/hbm_calib $ROOT/tools/hbm_calib.hip; }
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --batch $BATCH --steps $STEPS --warmup 0 --no-extras --no-cpu-baseline"
run_pass () { # name, counters..., -- command
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=($1); shift; done; shift
  timeout -k 5 600 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run_pass calib_fetch FETCH_SIZE -- $ROOT/tools/bin/hbm_calib
run_pass calib_write WRITE_SIZE -- $ROOT/tools/bin/hbm_calib
run_pass fetch FETCH_SIZE -- $BENCH
run_pass write WRITE_SIZE -- $BENCH
run_pass mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- $BENCH
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- $BENCH
run_pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -- $BENCH
cd $ROOT
for p in fetch write mfma tcc sq; do grep '^{' $OUT/$p.log | tail -1 > $OUT/$p.bench.json; done
python - "$OUT" "$TAG" "$BATCH" "$STEPS" <<'PY'
import csv, glob, json, sys, collections, subprocess, os
out, tag, batch, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
def load(name):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = ("scvx_persistent_kernel" if "scvx_persistent_kernel" in k else "ipm_kernel" if "ipm_kernel" in k else "discretize_kernel" if "discretize_kernel" in k else
                 "scvx_cost_update_kernel" if "scvx_cost_update" in k else "refill" if "refill" in k else
                 k.split("(")[0].split("::")[-1][:40])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "TCC_HIT_sum"):
                cnt[k] += 1
    return acc, cnt
def bench(name):
    try:
        return json.load(open(f"{out}/{name}.bench.json"))
    except Exception:
        return None
S = {"tag": tag, "batch": batch, "steps": steps, "command": f"bench.py --batch {batch} --steps {steps} --warmup 0 --no-extras --no-cpu-baseline under rocprofv3 --pmc (one pass per counter group)"}
try:
    S["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    S["commit"] = os.environ.get("GRAFT_COMMIT", "worktree")
try:
    sys.path.insert(0, "tools"); import csrc_hash; S["csrc_sha"] = csrc_hash.csrc_sha()  # identity of the kernel sources (bench.py: stale check)
except Exception as e:
    S["csrc_sha"] = "unknown: %s" % e
# ---- calibration ----
cf, _ = load("calib_fetch"); cw, _ = load("calib_write")
B4 = float(4 << 30); R50 = (B4 / 8 // 50) * 400
cal = {}
for k, known in (("calib_read8", B4), ("calib_read16", B4), ("calib_read8_rows50", R50)):
    if k in cf: cal[k] = {"known_bytes": known, "FETCH_SIZE_KB": cf[k]["FETCH_SIZE"], "counter_bytes_over_known": cf[k]["FETCH_SIZE"] * 1024 / known}
for k, known in (("calib_write8", B4), ("calib_write8_rows50", R50)):
    if k in cw: cal[k] = {"known_bytes": known, "WRITE_SIZE_KB": cw[k]["WRITE_SIZE"], "counter_bytes_over_known": cw[k]["WRITE_SIZE"] * 1024 / known}
S["calibration_kernels"] = cal
rf = cal.get("calib_read8_rows50", cal.get("calib_read8", {})).get("counter_bytes_over_known", 1.0) or 1.0
wf = cal.get("calib_write8_rows50", cal.get("calib_write8", {})).get("counter_bytes_over_known", 1.0) or 1.0
S["calibration"] = f"FETCH_SIZE x{1/rf:.3f}, WRITE_SIZE x{1/wf:.3f} (8 B/lane 400-byte rows streamed once over 4 GiB, same session)"
# ---- traffic ----
af, nf = load("fetch"); aw, nw = load("write")
bf, bw = bench("fetch"), bench("write")
K = {}
for k in ("scvx_persistent_kernel", "ipm_kernel", "discretize_kernel", "scvx_cost_update_kernel"):
    if k not in af and k not in aw:
        continue
    K[k] = {"dispatches": nf.get(k, 0), "fetch_bytes_raw": af[k]["FETCH_SIZE"] * 1024, "write_bytes_raw": aw[k]["WRITE_SIZE"] * 1024,
            "fetch_bytes": af[k]["FETCH_SIZE"] * 1024 / rf, "write_bytes": aw[k]["WRITE_SIZE"] * 1024 / wf}
S["kernels"] = K
if bf:
    inst = bf["config"]["instances_timed"]; it = bf["config"]["mean_ipm_iterations_per_trajectory"] * inst
    solves = bf["config"]["mean_subproblem_solves"] * inst; calls = bf["config"]["mean_scvx_iterations"] * inst
    # the kernel that carries the interior-point solves: the persistent SCvx kernel (one launch per job: refill + multipleShooting + solve + cost
    # of every instance, the default engine since round 5) or ipm_kernel (pool engine)
    main = "scvx_persistent_kernel" if "scvx_persistent_kernel" in K else "ipm_kernel"
    S["engine"] = "persistent" if main == "scvx_persistent_kernel" else "pools"
    tot = K[main]["fetch_bytes"] + K[main]["write_bytes"]
    S["ipm_iterations"] = it; S["ipm_solves"] = solves; S["trajectories"] = inst
    S["ipm_bytes_per_instance_iteration"] = tot / it  # (persistent engine: ALL steps' bytes over the interior-point iterations; the other steps move < 1 %)
    S["ipm_bytes_per_instance_iteration_raw"] = (K[main]["fetch_bytes_raw"] + K[main]["write_bytes_raw"]) / it
    S["ipm_bytes_per_instance_solve"] = tot / solves
    S["bytes_per_trajectory"] = tot / inst
    if "discretize_kernel" in K:
        S["discretize_bytes_per_instance_call"] = (K["discretize_kernel"]["fetch_bytes"] + K["discretize_kernel"]["write_bytes"]) / calls
        S["discretize_write_bytes_per_instance_call"] = K["discretize_kernel"]["write_bytes"] / calls
# ---- matrix core ----
am, _ = load("mfma")
mk = "scvx_persistent_kernel" if "scvx_persistent_kernel" in am else "ipm_kernel"
if mk in am:
    d = am[mk]; S["ipm_mfma"] = dict(d)
    bm = bench("mfma")
    if bm and d.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
        it = bm["config"]["mean_ipm_iterations_per_trajectory"] * bm["config"]["instances_timed"]
        S["ipm_mfma"]["MOPS_F64_per_instance_iteration"] = d["SQ_INSTS_VALU_MFMA_MOPS_F64"] / it
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES") and d.get("SQ_BUSY_CU_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the SIMDs (4 per CU, each with its own matrix pipe), SQ_BUSY_CU_CYCLES over the CUs: the fraction of
        # the time a matrix pipe is busy is busy / (4 x CU-busy).  (Until round 5 this file divided by the CU cycles alone and DESIGN quoted the
        # result, 0.60, as "matrix core busy 60 %": 4 x too high -- VERDICT r5 weak 7.)
        S["ipm_mfma"]["mfma_pipe_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"])
    if bm and d.get("SQ_INSTS_VALU_MFMA_MOPS_F64") and bm.get("roofline", {}).get("kernel_time_s"):
        # executed matrix-core flops: the counter is in units of 512 flop; over the kernel time of the SAME (profiled) run
        tf = d["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / bm["roofline"]["kernel_time_s"] / 1e12
        S["ipm_mfma"]["executed_mfma_TFLOPs"] = tf
        S["ipm_mfma"]["executed_mfma_frac_of_fp64_peak"] = tf / 78.6
        S["ipm_mfma"]["kernel_time_s_of_that_pass"] = bm["roofline"]["kernel_time_s"]
for name in ("tcc", "sq"):
    a, _ = load(name)
    S[name] = {k: dict(v) for k, v in a.items() if k in ("scvx_persistent_kernel", "ipm_kernel", "discretize_kernel", "scvx_cost_update_kernel")}
mk = "scvx_persistent_kernel" if "scvx_persistent_kernel" in S.get("tcc", {}) else "ipm_kernel"
if mk in S.get("tcc", {}):
    t = S["tcc"][mk]
    if t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0) > 0:
        S["ipm_l2_hit_rate"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
json.dump(S, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps({k: S[k] for k in S if k not in ("tcc", "sq")}, indent=1)[:6000])
PY
for p in calib_fetch calib_write fetch write mfma tcc sq; do rm -rf $OUT/$p; done
tail -3 $OUT/mfma.log | cut -c1-300
