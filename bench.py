#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X batched Successive-Convexification engine.

Metric (BASELINE.json): CONVERGED SCvx trajectories/sec, RocketQuat, K=50, on 1/2/4/8 MI355X.

One "step" = one batch of `--batch` (8192, BASELINE configs[2]) randomised RocketQuat instances PER GPU pushed through
SCvxAlgorithm::solve (cold start; shipped SCvx.info, K=50): per instance up to 30 SCvx iterations of
multipleShooting + SOCP solve + nonlinear cost + accept/reject (SCvxAlgorithm.cpp:61-164).  The `--steps` batches of the
timed region form ONE job of steps x batch instances per GPU for the streaming engine (continuous batching over `--batch`
resident slots, scpp_hip_scvx_solve_stream): a slot whose instance has terminated is refilled with the next queued instance
on the device, so the batches overlap instead of each draining through its own long tail (instances need 10..90 solves).
Timed: upload of the initial states -> on-device solve of every instance -> (N > 1) RCCL all-gather of the result rows ->
download of this rank's rows to the host.  `value` = instances that MET THE CONVERGENCE TEST |dL| < change_threshold
(SCvxAlgorithm.cpp:125) summed over ranks / wall time (max over ranks).  Weak scaling: per-GPU work fixed.

Secondary numbers (rank 0, outside the timed region) under `config`: the SC-mode (SCAlgorithm, what SC_oneshot runs) rate in
trajectories TERMINATED per second -- with the shipped weights none of them meets SCAlgorithm's convergence test at K=50,
see DESIGN.md -- an isolated single-batch SCvx solve, un-overlapped (single-pool) kernel times, and the linear-MPC leg.

Usage: python bench.py --gpus N --steps K --warmup W.  N > 1 without a launcher: the script starts its N ranks itself (one process per
GPU under torch.distributed.run on 127.0.0.1, `_self_launch`) and refuses when fewer than N GPUs are visible; under an external
torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  --backend gloo + --library <emulation build> drives the same N > 1
code path on CPU in tests/test_bench_distributed.py, with and without the launcher.
"""
import argparse
import glob
import re
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic work per unit
FLOP_PER_IPM_ITER = 2.0e6        # one Mehrotra iteration of one instance: 1 factorisation + 2 solves + cone algebra
FLOP_PER_SOCP_INIT = 1.6e6       # W=I factorisation + 2 solves + border (cold start: once per instance)
IPM_ALGO_BYTES_PER_SOLVE = 131712 + 7208 + 7200  # read dd + td, write X, U: what one sub-problem solve must move
DISC_BYTES_PER_INSTANCE = 139000  # 7,288 B read + 131,712 B written per instance-call
# One evaluation of the augmented right-hand side of the headline mode (first-order hold, fixed final time: V = [x | Phi | Psi_B | Psi_C], 14 x 23),
# counted from what discretizeSegment executes USEFULLY (14-wide, not the 16-wide tiles the matrix core multiplies):
#   A (14 x 14) times the 22 sensitivity columns            14 * 14 * 22 * 2 = 8 624
#   input forcing of Psi_B, Psi_C (B scaled by (dt-t)/dt, t/dt)  2 * 14 * 4 * 2 =   224
#   flow map + the 61 structural non-zeros of its Jacobian     ~   450
#   Runge-Kutta combination: 55 non-zero a_ij + 7 non-zero b_j of RKF78 on 14 * 23 entries, per stage (62 * 322 * 2 / 13)   3 071
# = 12.4 kflop.  (Rounds 3 - 5 priced it at 18.6 k, the 16-wide tile count; SURVEY 8(d)'s 19.7 k is the reference's Phi^-1 formulation.)
DISC_FLOP_PER_RHS = 12.4e3
# getNonlinearCost of one candidate (SCvxAlgorithm.cpp:262-278): 49 segments x 20 RKF78 steps x 13 stages = 12 740 flow-map evaluations of ~120 flop
# + their share of the stage combination (62 * 14 * 2 / 13 = 134 flop per stage) = 12 740 x 254
COST_FLOP_PER_SOLVE = 3.2e6
DISC_MAX_STEP = 12.0 / (14.0 * 5.0)  # csrc/discretize_kernel.h, opt-in rule: n = clamp(ceil(segment seconds / this), 1, 5) RKF78 steps per segment
PEAK_FP64_TFLOPS = 78.6           # MI355X FP64 vector == FP64 matrix peak (spec)
PEAK_HBM_GBS = 8000.0


def DISC_RULE_STEPS(model, K):
    import math

    return min(5, max(1, math.ceil(float(model.p.final_time) / (K - 1) / DISC_MAX_STEP)))


class _DevArray:
    """__cuda_array_interface__ view of a raw device pointer (zero-copy hand-over to torch for RCCL)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _scvx_cpu(K, seed, solver, seconds_budget, threads):
    """oracle SCvx runs (CPU restatement) on `threads` native host threads (oracle/capi.cpp: oracle_scvx_run_batch -- one counter,
    one std::thread per core, no interpreter between the instances).  The sample is sized from one timed instance so that the whole
    leg takes about `seconds_budget` seconds."""
    import ctypes as C

    import oracle_lib

    lib = oracle_lib.lib()
    cfg = oracle_lib.CONFIG_ROOT.encode()

    def batch(first, n, nthreads):
        counts = (C.c_longlong * 4)()
        t0 = time.time()
        rc = lib.oracle_scvx_run_batch(cfg, int(K), C.c_ulonglong(seed), C.c_ulonglong(first), int(n), int(solver), int(nthreads), counts)
        assert rc == 0
        return time.time() - t0, [int(v) for v in counts]

    t1, c1 = batch(0, 1, 1)
    n = int(max(threads, min(64 * threads, (seconds_budget / max(t1, 1e-3)) * threads)))
    dt, c = batch(0, n, threads)
    return dict(n=n, dt=dt, converged=c[0], failures=c[1], latency=t1, mean_iters=c[2] / n, mean_solves=c[3] / n, first=c1)


def _cgroup_cpu_quota():
    """CPUs the cgroup of this process may use per scheduling period (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us); None if
    unlimited or unreadable"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except Exception:
        return None


def cpu_baseline(K, seed, seconds_budget=12.0):
    """The oracle (CPU restatement of SCpp's algorithm, g++ -O2 -- NOT ECOS: the reference cannot be built)
    on the GPU box's host cores, same workload as the headline: converged SCvx trajectories/s."""
    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    quota = _cgroup_cpu_quota()
    # one native thread per CPU this process can actually use: the affinity mask, capped by the cgroup CPU quota of the box (the test
    # boxes show 256 logical CPUs under a 16-CPU quota: beyond 16 threads the throughput FALLS -- 27.8 /s at 16, 12.9 /s at 256,
    # tools/cpu_probe.py -- because the scheduler throttles the whole group)
    threads = max(1, min(usable, int(quota + 0.5)) if quota else usable)
    tw = _scvx_cpu(K, seed, 1, seconds_budget, threads)
    out = {
        "value": tw["converged"] / tw["dt"],
        "unit": "converged SCvx trajectories/s",
        "cores": threads,
        "host_logical_cpus": cores,       # os.cpu_count() of the GPU box
        "host_usable_cpus": usable,       # sched_getaffinity: what this process may run on
        "host_cgroup_cpu_quota": quota,   # cpu.max quota / period of the box's cgroup (None: unlimited)
        "kind": "port",
        "single_thread_latency_s": tw["latency"],
        "sample": f"{tw['n']} RocketQuat K={K} SCvx instances (seed {seed}, instances 0..{tw['n'] - 1}), oracle structured-IPM twin "
                  f"(CPU restatement of SCpp's algorithm -- not ECOS), g++ -O2, {threads} native threads (one per CPU of the cgroup quota / affinity mask); {tw['converged']} converged, "
                  f"{tw['failures']} solver failures, mean {tw['mean_iters']:.1f} SCvx iterations / {tw['mean_solves']:.1f} solves",
    }
    # the reference-shaped form: Epigraph-style literal problem (n=2273, p=814, m=2573) on the sparse ECOS restatement
    try:
        lt = _scvx_cpu(K, seed, 0, seconds_budget, threads)
        out["literal_form"] = {
            "value": lt["converged"] / lt["dt"], "unit": "converged SCvx trajectories/s", "cores": threads,
            "single_thread_latency_s": lt["latency"],
            "sample": f"{lt['n']} instances, oracle literal sparse-KKT solver (socp.hpp) on the reference-shaped problem; {lt['converged']} converged, "
                      f"{lt['failures']} solver failures, mean {lt['mean_iters']:.1f} SCvx iterations / {lt['mean_solves']:.1f} solves",
        }
    except Exception as e:
        out["literal_form"] = {"error": str(e)}
    return out


def measured_traffic():
    """HBM bytes per instance-IPM-iteration of ipm_kernel from the newest committed PMC summary (tools/pmc_hbm.sh ->
    profiles/r*_pmc_hbm_*.json); None if there is none."""
    best = None

    def version(f):
        m = re.search(r"r(\d+)_pmc_hbm_v(\d+)([a-z]?)", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2)), m.group(3)) if m else (-1, -1, "")

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_v*.json")), key=version):
        try:
            d = json.load(open(f))
            if "ipm_bytes_per_instance_iteration" in d:
                best = (f, d)
        except Exception:
            pass
    return best


def parity_summary():
    """Counts of the at-scale parity test of the headline mode (tests/test_gpu_parity.py::test_scvx_at_scale_parity_and_literal_audit writes
    gpurun_out/r06_parity_at_scale.json; the committed copy under profiles/ is what is reported here, with the kernel-source hash it was
    taken on): identical records, instances beyond 1e-5 in states / inputs, certified instances."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_at_scale*.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception as e:
        return {"error": str(e)}
    keep = ("instances", "identical_records", "non_identical_records", "non_identical_explained_by_a_tie", "largest_tie_margin",
            "instances_beyond_1e5_states", "instances_beyond_1e5_inputs", "flagged", "certified", "rel_dX", "rel_dU", "bars", "csrc_sha")
    out = {k: d[k] for k in keep if k in d}
    out["imported_from"] = os.path.relpath(files[-1], ROOT)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import csrc_hash
        out["stale"] = d.get("csrc_sha") != csrc_hash.csrc_sha()
    except Exception:
        out["stale"] = None
    return out


def _self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-run this very command line as N ranks (one process per GPU) under
    torch.distributed.run on the loopback interface; the children see WORLD_SIZE / RANK / LOCAL_RANK and take the distributed path of
    main(), rank 0 prints the JSON line to the inherited stdout.  Fails before anything is started when the node has fewer GPUs."""
    import socket
    import subprocess

    if args.backend != "gloo":
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible; one rank per GPU, no oversubscription")
    with socket.socket() as s:  # a free loopback port for the rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8192, help="instances per step and per GPU = resident slots (BASELINE configs[2]: 8192)")
    ap.add_argument("--K", type=int, default=50)
    ap.add_argument("--seed", type=int, default=20260927)
    ap.add_argument("--pools", type=int, default=0, help="slot pools (HIP streams) of the streaming engine; 0 = library default")
    ap.add_argument("--max-iterations", type=int, default=None, help="override SCvx.info max_iterations (tests)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for the CPU test of this script)")
    ap.add_argument("--library", default=None, help="path of the C-ABI library (tests pass the CPU emulation build)")
    ap.add_argument("--dump", default=None, help="rank 0 writes the gathered result rows to this .npy (tests)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the multi-GPU result path (zero-copy view of the library's rows -> torch staging tensor -> "
                         "all_gather_into_tensor, in chunks) inside the timed region even with ONE rank: a world-1 process group "
                         "of --backend is created, so a 1-GPU box executes the RCCL branch")
    ap.add_argument("--gather-chunk-mb", type=float, default=64.0, help="largest per-rank payload of one all-gather (MB)")
    ap.add_argument("--placement-candidates", type=int, default=None,
                    help="contexts allocated side by side, a short probe job on each, the best one kept (SCvxAlgorithm.initialize; DESIGN.md 5: the placement "
                         "regimes; measured in round 6: no gain on a box whose first allocation is already in the fast regime, profiles/r06_placement_regimes.json).  Default 1 = off.  Outside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (SC mode, single batch, single pool, MPC)")
    ap.add_argument("--mpc-batch", type=int, default=32768)
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run, loopback
        # rendezvous on a free port) and hand its exit code back.  Under an external launcher WORLD_SIZE is set and this is skipped.
        return _self_launch(args)

    import torch
    import scpp_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a launcher that started a different number of ranks than --gpus names would silently mislabel the line
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or omit the launcher)")
    if args.backend != "gloo" and torch.cuda.device_count() < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible; one rank per GPU, no oversubscription")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = args.backend != "gloo"
    dist = None
    use_dist = world > 1 or args.force_gather
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if on_gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    elif on_gpu:
        torch.cuda.set_device(0)
    dev_index = (local_rank if world > 1 else 0) if on_gpu else 0
    dev = torch.device("cuda", dev_index) if on_gpu else torch.device("cpu")

    B, K = args.batch, args.K
    model = scpp_amd.RocketQuat().loadParameters()
    n_place = args.placement_candidates if args.placement_candidates is not None else 1
    alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=B, device=dev_index, library=args.library,
                                 max_iterations=args.max_iterations).initialize(placement_candidates=n_place)
    ctx = alg.ctx
    rowd = K * 18 + len(ctx.STREAM_SCALARS)

    def barrier():
        if use_dist:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def states(first_step, nsteps):
        # instance ids are disjoint across ranks and steps (fresh problems every step)
        return np.concatenate([model.randomized_initial_states(B, seed=args.seed, first=((first_step + i) * world + rank) * B)
                               for i in range(nsteps)], axis=0)

    def run_job(x0):
        """one streaming job + the collective + the download; returns (#converged on this rank, local rows, gathered rows)"""
        nconv = alg.solveStream(x0, slots=B, pools=args.pools)
        n = x0.shape[0]
        gathered = None
        if use_dist:
            ptr, rd, nrows = ctx.stream_rows_device()
            assert rd == rowd and nrows == n
            if on_gpu:
                # zero-copy view of the library's result rows (hipMalloc'ed memory torch's allocator does not own)
                src = torch.as_tensor(_DevArray(ptr, (n, rowd)), device=dev)
            else:
                src = torch.from_numpy(ctx.stream_download_rows())
            # One collective per chunk of <= --gather-chunk-mb per rank: the receive buffer of a collective is world x chunk
            # (512 MB on 8 GPUs at the default) instead of the whole job at once, and the staging copy of chunk i+1 (torch's
            # stream) is independent of the collective of chunk i.  Every rank holds the same n, so the chunking is identical
            # everywhere.  Rank r's rows of chunk c land at gathered[r * n + lo : r * n + hi].
            rows_per = max(1, int(args.gather_chunk_mb * 1e6) // (rowd * 8))
            gathered = torch.empty((world * n, rowd), dtype=torch.float64, device=dev)
            gview = gathered.view(world, n, rowd)
            n_coll = 0
            for lo in range(0, n, rows_per):
                hi = min(n, lo + rows_per)
                mine = torch.empty((hi - lo, rowd), dtype=torch.float64, device=dev)  # torch-owned staging tensor
                mine.copy_(src[lo:hi])
                recv = torch.empty((world * (hi - lo), rowd), dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(recv, mine)
                gview[:, lo:hi, :] = recv.view(world, hi - lo, rowd)
                n_coll += 1
            gather_stats["collectives"] = n_coll
            gather_stats["rows_per_collective"] = rows_per
            gather_stats["bytes_per_rank_per_collective"] = min(n, rows_per) * rowd * 8
        out = ctx.stream_download()  # D2H of this rank's rows: getSolution is part of the path's contract
        return nconv, out, gathered

    gather_stats = {}
    x_warm = states(0, args.warmup) if args.warmup > 0 else None
    x_timed = states(args.warmup, args.steps)  # host-side generation outside the timed region
    if x_warm is not None:
        run_job(x_warm)
    ctx.timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    nconv, out, gathered = run_job(x_timed)
    barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timing(reset=True)
    rounds = ctx.stream_rounds() if hasattr(ctx, "stream_rounds") else None
    stream_profile = ctx.stream_profile()  # per-step wavefront ticks of the persistent engine (zeros after a pool-engine job)

    total = x_timed.shape[0]
    local = np.array([nconv, total, int(out["sc_iters"].sum()), int(out["solves"].sum()), int(out["ipm_iters"].sum()),
                      int((out["status"] != 0).sum())], dtype=np.float64)
    assert (out["instance"] == np.arange(total)).all(), "result rows out of order"
    assert int(out["converged"].sum()) == nconv
    if use_dist:
        # the gathered rows of this rank are, bitwise, the rows the library holds (the zero-copy view, the staging copy and the
        # collective moved bytes, nothing else)
        local_rows = ctx.stream_download_rows()
        mine_back = gathered.view(world, total, rowd)[rank].cpu().numpy()
        assert np.array_equal(mine_back.view(np.uint64), local_rows.view(np.uint64)), "gathered rows differ from the library's rows"
        gather_stats["gathered_equals_local_bitwise"] = True
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        agg = torch.tensor(local, dtype=torch.float64, device=dev)
        dist.all_reduce(agg)
        g_conv, g_total, g_iters, g_solves, g_ipm, g_fail = [float(v) for v in agg.tolist()]
        # the gathered rows carry the same story as the reduced counters (every rank holds every trajectory)
        conv_col = K * 18 + ctx.STREAM_SCALARS.index("converged")
        assert int(round(float(gathered[:, conv_col].sum().item()))) == int(g_conv), "all-gather payload disagrees with the counters"
        if rank == 0 and args.dump:
            np.save(args.dump, gathered.cpu().numpy())
    else:
        g_conv, g_total, g_iters, g_solves, g_ipm, g_fail = [float(v) for v in local]
        if args.dump:
            np.save(args.dump, gathered.cpu().numpy() if gathered is not None else ctx.stream_download_rows())

    extras = {}
    if rank == 0:
        # ---- what "converged" buys (north_star: "final virtual-control norm reported").  The reference's test is |dL| < change_threshold
        # (SCvxAlgorithm.cpp:125) and nothing else: it does NOT ask for nu -> 0, and a shrinking trust radius makes dL small.  Reported: how many converged
        # runs also meet the SC mode's nu_tol (1e-5), where the radius ended, and -- on the product's own simulate kernel, in SI units -- how far the
        # nonlinear propagation of a returned trajectory lands from its own next node. ----
        try:
            cv = out["converged"] == 1
            nu_c = out["nu_norm"][cv]
            tr_c = out["trust_region"][cv]
            meaning = {
                "convergence_test": "|dL| < change_threshold = 1e-3 (SCvxAlgorithm.cpp:125), the reference's only test",
                "converged_with_final_nu_norm1_below_1e-5": float((nu_c < 1e-5).mean()) if nu_c.size else None,
                "final_nu_norm1_percentiles_5_50_95": [float(v) for v in np.percentile(nu_c, [5, 50, 95])] if nu_c.size else None,
                "final_trust_radius_percentiles_5_50_95": [float(v) for v in np.percentile(tr_c, [5, 50, 95])] if tr_c.size else None,
                "initial_trust_radius": float(alg.opts.trust_region),
                "median_scvx_iterations_of_converged": float(np.median(out["sc_iters"][cv])) if cv.any() else None,
            }
            ns = int(min(128, cv.sum(), max(1, B // (K - 1))))
            if ns > 0 and on_gpu:
                idx = np.flatnonzero(cv)[:ns]
                Xs, Us = out["X"][idx], out["U"][idx]           # dimensional (SI) trajectories as returned
                sim = scpp_amd.Context(K=K, batch_max=ns * (K - 1), device=dev_index, library=args.library)
                sim.set_flow_params(np.tile(model.flow_params(nondimensionalize=False), (ns * (K - 1), 1)))
                dts = np.repeat(out["sigma"][idx] / (K - 1), K - 1)
                xp = sim.simulate(dts, Us[:, :-1].reshape(-1, 4), Us[:, 1:].reshape(-1, 4), Xs[:, :-1].reshape(-1, 14)).reshape(ns, K - 1, 14)
                sim.close()
                d = xp - Xs[:, 1:]
                pos = np.linalg.norm(d[:, :, 1:4], axis=2).max(axis=1)
                vel = np.linalg.norm(d[:, :, 4:7], axis=2).max(axis=1)
                meaning["dimensional_defect_of_returned_trajectories"] = {
                    "sample": f"the first {ns} converged instances, every segment propagated with scpp_hip_simulate (RKF78 x 20, first-order hold) in SI units",
                    "median_over_instances_of_max_position_defect_m": float(np.median(pos)),
                    "p95_max_position_defect_m": float(np.percentile(pos, 95)),
                    "median_over_instances_of_max_velocity_defect_m_per_s": float(np.median(vel)),
                    "median_descent_distance_m": float(np.median(np.linalg.norm(Xs[:, 0, 1:4], axis=1))),
                }
            extras["what_converged_means"] = meaning
        except Exception as e:
            extras["what_converged_means"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_extras:  # (single-GPU runs only: at N > 1 the other ranks would idle behind these legs)
        # ---- un-overlapped kernel times: the same engine with ONE slot pool (one stream), a 2-batch job ----
        try:
            ctx.timing(reset=True)
            xs = model.randomized_initial_states(2 * B, seed=args.seed, first=20_000_000)
            ts0 = time.perf_counter()
            nc1 = alg.solveStream(xs, slots=B, pools=1)
            o1 = ctx.stream_download()
            ts = time.perf_counter() - ts0
            t1 = ctx.timing(reset=True)
            ipm_s = t1["ms_socp"] * 1e-3
            fl = float(o1["ipm_iters"].sum()) * FLOP_PER_IPM_ITER + 2 * B * FLOP_PER_SOCP_INIT
            extras["single_pool"] = {
                "note": "one stream: hipEvent spans of consecutive kernels do not overlap",
                "converged_trajectories_per_s": nc1 / ts,
                "ipm_kernel_avg_launch_ms": t1["ms_socp"] / max(t1["n_socp"], 1),
                "ipm_kernel_launches": t1["n_socp"],
                "ipm_kernel_fp64_TFLOPs": fl / ipm_s / 1e12 if ipm_s > 0 else None,
                "ipm_kernel_fp64_frac": fl / ipm_s / 1e12 / PEAK_FP64_TFLOPS if ipm_s > 0 else None,
                "ipm_share_of_wall": ipm_s / ts,
                "discretize_kernel_avg_launch_ms": t1["ms_discretize"] / max(t1["n_discretize"], 1),
                "discretize_share_of_wall": t1["ms_discretize"] * 1e-3 / ts,
            }
        except Exception as e:
            extras["single_pool"] = {"error": str(e)}
        # ---- the opt-in step-length rule of discretize_kernel (round 3's default; since round 4 the default is the reference's five RKF78
        # steps per segment and the HEADLINE is measured with it): 2 steps at K = 50, 1e-13 away in A .. z, same engine and
        # pools as the headline, a 2-batch job ----
        try:
            ctx.set_discretization_steps(0)
            xs = model.randomized_initial_states(2 * B, seed=args.seed, first=30_000_000)
            tp0 = time.perf_counter()
            nc5 = alg.solveStream(xs, slots=B, pools=args.pools)
            o5 = ctx.stream_download()
            tp = time.perf_counter() - tp0
            extras["step_length_rule"] = {
                "note": "scpp_hip_set_discretization_steps(ctx, 0): n = clamp(ceil(segment seconds / 0.1714 s), 1, 5) RKF78 steps per segment "
                        "instead of the reference's fixed five (discretizationImplementation.hpp:141,154), which the headline uses; 2 batches "
                        "as one streaming job.  NOT the reference's scheme: a handful of accept / reject decisions differ",
                "rkf78_steps_per_segment": DISC_RULE_STEPS(model, K), "converged_trajectories_per_s": nc5 / tp, "converged_fraction": nc5 / (2 * B),
                "mean_subproblem_solves": float(o5["solves"].mean()),
            }
        except Exception as e:
            extras["step_length_rule"] = {"error": str(e)}
        finally:
            ctx.set_discretization_steps(5)
        # ---- one isolated batch through the plain batch entry point (latency view: includes its own tail) ----
        try:
            xv = model.randomized_initial_states(B, seed=args.seed, first=10_000_000)
            tv0 = time.perf_counter()
            nconv_v = alg.solve(xv)
            vout = alg.getSolution()
            tv = time.perf_counter() - tv0
            extras["single_batch"] = {
                "entry": "scpp_hip_scvx_solve: one batch, no refill", "batch": int(B),
                "converged_trajectories_per_s": nconv_v / tv, "converged_fraction": nconv_v / B,
                "max_subproblem_solves": int(vout["solves"].max()), "mean_subproblem_solves": float(vout["solves"].mean()),
            }
        except Exception as e:
            extras["single_batch"] = {"error": str(e)}
        ctx.close()
        # ---- the other instantiations of the persistent kernel (round 6: both models, both input holds): a 2-batch streaming job each ----
        try:
            import shutil, tempfile

            legs = {}
            # (+ the third model, Lander3dof: registered through the plugin list, pool engine -- csrc/sc_kernels.h; not a model of the reference)
            for mdl, foh, Kx in (("RocketQuat", False, K), ("Rocket2D", True, 30), ("Rocket2D", False, 30), ("Lander3dof", True, 30)):
                d = tempfile.mkdtemp()
                cfg = os.path.join(d, "config")
                shutil.copytree(os.path.join(os.path.dirname(scpp_amd.__file__), "config"), cfg)
                pth = os.path.join(cfg, mdl, "SCvx.info")
                t = open(pth).read()
                if not foh:
                    t = t.replace("interpolate_input                   true", "interpolate_input                   false")
                if mdl == "Rocket2D":  # the shipped file runs in SI units and does not converge; nondimensionalised it does
                    t = t.replace("nondimensionalize                   false", "nondimensionalize                   true")
                open(pth, "w").write(t)
                mx = {"RocketQuat": scpp_amd.RocketQuat, "Rocket2D": scpp_amd.Rocket2D, "Lander3dof": scpp_amd.Lander3dof}[mdl](cfg).loadParameters()
                ax = scpp_amd.SCvxAlgorithm(mx, K=Kx, batch_max=B, device=dev_index, library=args.library).initialize()
                xx = mx.randomized_initial_states(2 * B, seed=args.seed, first=40_000_000)
                ax.solveStream(xx[:256], slots=256)
                tx0 = time.perf_counter()
                ncx = ax.solveStream(xx, slots=B)
                ox = ax.ctx.stream_download()
                tx = time.perf_counter() - tx0
                legs["%s_%s_K%d" % (mdl, "FOH" if foh else "ZOH", Kx)] = {
                    "engine": "persistent kernel" if ax.ctx.stream_rounds()["pools"] == 0 else "pool engine",
                    "converged_trajectories_per_s": ncx / tx, "converged_fraction": ncx / (2 * B), "solver_failures": int((ox["status"] != 0).sum()),
                    "mean_subproblem_solves": float(ox["solves"].mean()), "mean_ipm_iterations_per_trajectory": float(ox["ipm_iters"].mean()),
                    "config": "SCvx.info with interpolate_input %s%s" % ("true" if foh else "false", ", nondimensionalize true" if mdl == "Rocket2D" else ""),
                }
                ax.ctx.close()
                shutil.rmtree(d, ignore_errors=True)
            extras["other_instantiations"] = legs
        except Exception as e:
            extras["other_instantiations"] = {"error": str(e)}
        # ---- SC mode (SCAlgorithm: what SC_oneshot runs): terminated trajectories/s ----
        try:
            salg = scpp_amd.SCAlgorithm(model, K=K, batch_max=B, device=dev_index, library=args.library).initialize()
            xs = model.randomized_initial_states(2 * B, seed=args.seed, first=30_000_000)
            salg.solve(xs[:B])
            salg.ctx.synchronize()
            salg.ctx.timing(reset=True)
            ts0 = time.perf_counter()
            nconv_s = salg.solve(xs[B:])
            sout = salg.getSolution()
            ts = time.perf_counter() - ts0
            st = salg.ctx.timing(reset=True)
            extras["sc_mode"] = {
                "algorithm": "SCAlgorithm (SCAlgorithm.cpp:66-189), shipped SC.info weights, K=%d, cold start" % K,
                "batch": int(B), "terminated_trajectories_per_s": B / ts, "converged_trajectories_per_s": nconv_s / ts,
                "converged_fraction": nconv_s / B, "mean_sc_iterations": float(sout["sc_iters"].mean()),
                "mean_ipm_iterations_per_trajectory": float(sout["ipm_iters"].mean()),
                "solver_failures": int((sout["status"] != 0).sum()),
                "median_final_virtual_control_norm1": float(np.median(sout["nu_norm"])),
                "ipm_kernel_avg_launch_ms_two_stream_overlapped": st["ms_socp"] / max(st["n_socp"], 1),
            }
            salg.ctx.close()
        except Exception as e:
            extras["sc_mode"] = {"error": str(e)}
        # ---- linear MPC path (Rocket2D, MPCAlgorithm / MPC_sim) ----
        if args.mpc_batch > 0:
            try:
                m2 = scpp_amd.Rocket2D().loadParameters()
                m2.p.constrain_initial_final = False  # model.info: "enable for SC and disable for MPC/LQR"
                Bm = args.mpc_batch
                malg = scpp_amd.MPCAlgorithm(m2, batch_max=Bm, device=dev_index, library=args.library).initialize()
                xm = m2.randomized_initial_states(Bm, seed=args.seed)
                malg.setInitialState(xm); malg.setFinalState(m2.p.x_final)
                malg.solve()  # warm-up
                malg.ctx.timing(reset=True)
                tm0 = time.perf_counter()
                reps = 10
                for _ in range(reps):
                    nok = malg.solve()
                tmw = (time.perf_counter() - tm0) / reps
                mt = malg.ctx.timing(reset=True)
                mout = malg.getSolution()
                extras["mpc_mode"] = {
                    "algorithm": "MPCAlgorithm on the shipped Rocket2D MPC.info, K=%d, constant dynamics, cold start per solve" % malg.K,
                    "batch": int(Bm), "solves_per_s": Bm / tmw, "avg_launch_ms": mt["ms_socp"] / max(mt["n_socp"], 1),
                    "solved_fraction": nok / Bm, "mean_ipm_iterations": float(mout["iters"].mean()),
                }
                malg.ctx.close()
            except Exception as e:
                extras["mpc_mode"] = {"error": str(e)}

    if rank == 0:
        value = g_conv / dt
        # roofline of the dominant kernel (ipm_kernel), rank 0's launches of the timed region, hipEvent spans on the launching
        # streams.  With more than one slot pool the spans of different streams overlap in time (the kernels share the CUs), so
        # the per-launch duration is an upper bound of the kernel's own time; `single_pool` has the un-overlapped figure.
        ipm_iters0 = float(out["ipm_iters"].sum())
        socp_flops = ipm_iters0 * FLOP_PER_IPM_ITER + total * FLOP_PER_SOCP_INIT
        launches = max(tm["n_socp"], 1)
        # engine of the timed job: the persistent kernel (ONE launch: refill + multipleShooting + solve + cost of every instance; the library's
        # default for this configuration since round 5) reports no pools
        persistent = bool(rounds) and rounds.get("pools", 1) == 0
        disc_steps = 5  # the library default since round 4: the reference's fixed count (scpp_hip_set_discretization_steps)
        DISC_FLOP_PER_INSTANCE = (K - 1) * 13 * disc_steps * DISC_FLOP_PER_RHS
        disc_flops0 = float(out["sc_iters"].sum()) * DISC_FLOP_PER_INSTANCE
        cost_flops0 = float(out["solves"].sum()) * COST_FLOP_PER_SOLVE
        step_view = None
        if persistent:
            ticks = dict(stream_profile)
            tsum = sum(ticks.values()) or 1.0
            ksec = tm["ms_socp"] * 1e-3
            step_view = {"definition": "share = the step's wavefront time (s_memtime ticks summed over wavefronts) / all steps'; a step's TFLOP/s = its algorithmic "
                                       "flops / (kernel time x share): the rate it runs at while the chip's wavefront slots hold the measured mix of steps"}
            for name, fl in (("solve", socp_flops), ("discretize", disc_flops0), ("cost", cost_flops0), ("refill", 0.0)):
                sh = ticks.get(name, 0.0) / tsum
                step_view[name] = {"share_of_wavefront_time": sh, "flops": fl, "Mcycles_per_trajectory": ticks.get(name, 0.0) / max(total, 1) / 1e6,
                                   "fp64_TFLOPs_in_its_share": (fl / (ksec * sh) / 1e12) if (sh > 0 and ksec > 0) else None,
                                   "fp64_frac_in_its_share": (fl / (ksec * sh) / 1e12 / PEAK_FP64_TFLOPS) if (sh > 0 and ksec > 0) else None}
            socp_flops_all = socp_flops + disc_flops0 + cost_flops0
        else:
            socp_flops_all = socp_flops
        # kernel time = length of the UNION of the launches' hipEvent spans on a common time axis (launches of the slot pools
        # overlap in time: the plain sum of spans exceeds the wall clock).  By construction <= the timed region; asserted.
        span_sum_s = tm["ms_socp"] * 1e-3
        socp_s = tm.get("ms_socp_union", 0.0) * 1e-3 or span_sum_s
        assert socp_s <= dt * 1.001, f"ipm_kernel time {socp_s:.3f} s exceeds the timed region {dt:.3f} s"
        achieved_tf = socp_flops_all / socp_s / 1e12 if socp_s > 0 else 0.0
        achieved_tf_span_sum = socp_flops_all / span_sum_s / 1e12 if span_sum_s > 0 else 0.0
        # continuity with rounds 1 - 4: the interior-point flops alone over the same kernel time
        achieved_tf_solve_only = socp_flops / socp_s / 1e12 if socp_s > 0 else 0.0
        pmc = measured_traffic()
        traffic = None
        traffic_source = None
        traffic_note = "no profiles/r*_pmc_hbm_*.json summary found: traffic unmeasured"
        if pmc is not None:
            f, d = pmc
            traffic = d["ipm_bytes_per_instance_iteration"] * ipm_iters0 / launches
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import csrc_hash
                here = csrc_hash.csrc_sha()
            except Exception:
                here = None
            # stale: the PMC summary was taken from other kernel sources than the ones this run uses (content hash of scpp_amd/csrc +
            # include/scpp_hip.h; the GPU box has no .git to ask for ancestry)
            stale = (here is None) or (d.get("csrc_sha") != here)
            traffic_source = {"imported_from": os.path.relpath(f, ROOT), "commit_of_that_profile": d.get("commit", "?"),
                              "csrc_sha_of_that_profile": d.get("csrc_sha"), "csrc_sha_of_this_run": here, "stale": stale,
                              "measured_in_this_run": False}
            traffic_note = (f"IMPORTED, not measured in this run: {os.path.relpath(f, ROOT)} (rocprofv3 PMC FETCH_SIZE + WRITE_SIZE, separate "
                            f"passes, of ipm_kernel at commit {d.get('commit', '?')}, {d.get('calibration', 'uncalibrated')}) = "
                            f"{d['ipm_bytes_per_instance_iteration']:.3e} B per instance-IPM-iteration, scaled by this run's iterations "
                            f"per launch; algorithmic minimum (read dd + td, write X, U) = {IPM_ALGO_BYTES_PER_SOLVE} B per instance-solve")
        # matrix-core utilisation from the same imported PMC summary (tools/pmc_hbm.sh: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), and the
        # executed matrix-core flops SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 over that pass's kernel time): the two must agree (busy fraction x peak ~ executed)
        mfma_pipe = None
        if pmc is not None and isinstance(pmc[1].get("ipm_mfma"), dict):
            mm = pmc[1]["ipm_mfma"]
            mfma_pipe = {"mfma_pipe_busy_frac": mm.get("mfma_pipe_busy_frac"), "executed_mfma_TFLOPs": mm.get("executed_mfma_TFLOPs"),
                         "executed_mfma_frac_of_fp64_peak": mm.get("executed_mfma_frac_of_fp64_peak"),
                         "note": "imported with `traffic` (same summary, same staleness flag); executed = 16-wide tile flops incl. padding and the transposes' "
                                 "structural zeros, so it exceeds the algorithmic `achieved`"}
        measured_gbs = (traffic * launches / socp_s / 1e9) if (traffic is not None and socp_s > 0) else None
        mfma_frac = achieved_tf / PEAK_FP64_TFLOPS
        hbm_frac = measured_gbs / PEAK_HBM_GBS if measured_gbs is not None else None
        limiting = ("hbm traffic of the workspace (measured bytes): %.0f GB/s = %.1f %% of peak, against %.1f %% of the FP64 matrix peak"
                    % (measured_gbs, 100 * hbm_frac, 100 * mfma_frac)) if (hbm_frac is not None and hbm_frac > mfma_frac) else "fp64 mfma"
        disc_s = tm.get("ms_discretize_union", 0.0) * 1e-3 or tm["ms_discretize"] * 1e-3
        if persistent:  # no launches of its own: the step's share of the persistent kernel
            disc_s = tm["ms_socp"] * 1e-3 * step_view["discretize"]["share_of_wavefront_time"]
        # flops the kernel EXECUTES: 13 stages x n steps per segment (the reference's scheme is n = 5 whatever K: 5.9e7 flop at K = 50)
        # discretize launches are masked (needs_disc): instances that re-solve after a rejection skip it, so count solves
        line = {
            "metric": "converged SCvx trajectories/sec (RocketQuat, K=50) at 1/2/4/8 MI355X",
            "value": value,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"RocketQuat SCvx (SCvxAlgorithm: fixed final time, hard input trust region, FOH), K={K}, {args.steps} steps x "
                            f"batch={B} randomised initial states per GPU (BASELINE configs[2]/[4]: 8192 per GPU, 65536 on 8), shipped "
                            f"Falcon-9 model.info + SCvx.info" +
                            (f"; the literal one-shot batch of configs[2] (ONE batch of {B} through scpp_hip_scvx_solve, its own tail included): "
                             f"{extras['single_batch']['converged_trajectories_per_s']:.0f} converged/s (config.single_batch)"
                             if isinstance(extras.get("single_batch"), dict) and "converged_trajectories_per_s" in extras["single_batch"] else ""),
                "algorithm": "SCvxAlgorithm::solve, cold start per instance; converged = |dL| < change_threshold (SCvxAlgorithm.cpp:125); "
                             "multipleShooting with the reference's 5 RKF78 steps per segment (discretizationImplementation.hpp:141,154)",
                "engine": (f"scpp_hip_scvx_solve_stream, persistent kernel (csrc/scvx_persistent.h): ONE launch per job, a wavefront per slot ({B} slots, "
                           f"2048 resident) takes instance after instance through refill -> multipleShooting -> sub-problem solve -> cost / accept / reject; "
                           f"steps x batch instances queued") if persistent else
                          f"scpp_hip_scvx_solve_stream, pool engine: continuous batching over {B} resident slots per GPU in rounds of launches, steps x batch instances queued",
                "global_batch": int(B * world),
                "instances_timed": int(g_total),
                "parallelism": (f"instance-sharded x{world}, no collective in the loop, {args.backend} all-gather of the result rows "
                                f"({rowd * 8} B per instance) in chunks of <= {args.gather_chunk_mb:g} MB per rank") if use_dist else "single GPU",
                "gather": gather_stats if use_dist else None,
                "converged_fraction": g_conv / g_total if g_total else 0.0,
                "terminated_trajectories_per_s": g_total / dt,
                "mean_scvx_iterations": g_iters / g_total if g_total else 0.0,
                "mean_subproblem_solves": g_solves / g_total if g_total else 0.0,
                "mean_ipm_iterations_per_trajectory": g_ipm / g_total if g_total else 0.0,
                "interior_point": ("ECOS's algorithm (NT scaling, Mehrotra predictor-corrector, sigma = (1 - alpha_aff)^3, step-to-boundary 0.99, feastol 1e-8, abstol / reltol "
                                   "1e-7) on the structured system, with ONE departure since round 6: primal and dual step lengths of their own (csrc/ipm_solve.h: "
                                   "IPM_SPLIT_STEPS) -- same-box A/B against the common step length: 333.6 vs 367.9 iterations per trajectory, 6213 vs 5733 converged/s "
                                   "(profiles/r06_ab_split_steps.json)"),
                "solver_failures": g_fail,
                "median_final_virtual_control_norm1": float(np.median(out["nu_norm"])),
                "median_final_nonlinear_defect": float(np.median(out["nonlinear_cost"])),
                "rounds": rounds,
                "stream_profile_ticks": stream_profile,
                "parity": parity_summary(),
                "placement": ({"what": "SCvxAlgorithm.initialize(placement_candidates): candidate contexts allocated side by side, a short warm streaming job on "
                                       "each, the fastest kept, before the warm-up and outside the timed region (DESIGN.md 5: one library is 2.5 - 3.6 % faster or "
                                       "slower by where the driver placed a context's allocations)", **alg.placement} if getattr(alg, "placement", None) else
                              {"candidates": 1}),
                **extras,
            },
            "roofline": {
                "kernel": ("scvx_persistent_kernel (one launch per job; per instance and SCvx iteration: multipleShooting of 49 segments, the batched structured "
                           "interior-point solve with block factorisations on v_mfma_f64_16x16x4_f64, the nonlinear cost of the candidate)") if persistent else
                          "ipm_kernel (batched structured IPM, one wavefront per instance, block factorisations on v_mfma_f64_16x16x4_f64)",
                "steps": step_view,
                "bound": "mfma",  # the roof north_star prices the KKT factorisation against; see limiting_resource
                "achieved": achieved_tf,
                "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s",
                "frac": mfma_frac,
                "achieved_definition": ("ALL steps of the kernel (interior-point solve + multipleShooting + candidate cost: flop_model) over its duration"
                                        if persistent else "the interior-point solve's flops over the kernel time"),
                "achieved_solve_only": achieved_tf_solve_only,   # the interior-point flops alone (what rounds 1 - 4 reported as `achieved`)
                "frac_solve_only": achieved_tf_solve_only / PEAK_FP64_TFLOPS,
                "mfma_pipe": mfma_pipe,
                "limiting_resource": limiting,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "traffic_note": traffic_note,
                "kernel_time_s": socp_s,
                "kernel_time_definition": "hipEvent span of the one persistent launch" if persistent else
                                          "union of the launches' hipEvent spans (exclusive wall time with >= 1 ipm_kernel in flight)",
                "timed_region_s": dt,
                "exclusive_ms_per_launch": 1e3 * socp_s / launches,
                "avg_launch_ms": tm["ms_socp"] / launches,  # average hipEvent span of one launch = what rocprofv3 --stats averages
                "frac_of_span_sum": achieved_tf_span_sum / PEAK_FP64_TFLOPS,
                "launches": tm["n_socp"],
                "flop_model": f"{FLOP_PER_IPM_ITER:.2e} flop per IPM iteration x {ipm_iters0:.0f} measured IPM iterations + "
                              f"{FLOP_PER_SOCP_INIT:.2e} per cold start x {total} instances (rank 0)" +
                              (f" + {DISC_FLOP_PER_INSTANCE:.2e} per multipleShooting call x {float(out['sc_iters'].sum()):.0f} calls + "
                               f"{COST_FLOP_PER_SOLVE:.1e} per candidate cost x {float(out['solves'].sum()):.0f} solves" if persistent else ""),
                "hbm_view": {
                    "algorithmic_bytes_per_launch": IPM_ALGO_BYTES_PER_SOLVE * float(out["solves"].sum()) / launches,
                    "algorithmic_GBs": IPM_ALGO_BYTES_PER_SOLVE * float(out["solves"].sum()) / socp_s / 1e9 if socp_s > 0 else None,
                    "measured_GBs": measured_gbs,
                    "measured_frac": hbm_frac,
                    "peak_GBs": PEAK_HBM_GBS,
                },
            },
            "kernels": {
                "discretize": {
                    "kernel": "multipleShooting step of scvx_persistent_kernel (discretizeSegment<RocketQuat,FOH,fixed time>, the body of discretize_kernel)" if persistent
                              else "discretize_kernel<RocketQuat,FOH,fixed time>",
                    "avg_launch_ms": tm["ms_discretize"] / max(tm["n_discretize"], 1),
                    "launches": tm["n_discretize"],
                    "instance_calls": g_iters / world,
                    "kernel_time_s": disc_s,
                    "fp64_TFLOPs": (g_iters / world) * DISC_FLOP_PER_INSTANCE / disc_s / 1e12 if disc_s > 0 else None,
                    "fp64_frac": (g_iters / world) * DISC_FLOP_PER_INSTANCE / disc_s / 1e12 / PEAK_FP64_TFLOPS if disc_s > 0 else None,
                    "hbm_GBs": (g_iters / world) * DISC_BYTES_PER_INSTANCE / disc_s / 1e9 if disc_s > 0 else None,
                    "rkf78_steps_per_segment": disc_steps,
                    "flop_per_instance_call": DISC_FLOP_PER_INSTANCE,
                    "flop_per_instance_call_reference_scheme": (K - 1) * 65 * DISC_FLOP_PER_RHS,
                    "bound": "fp64-alu (170 .. 450 flop/B: HBM is not the binding roof, SURVEY §8(d))",
                },
            },
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(K, args.seed)
            except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
                line["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
