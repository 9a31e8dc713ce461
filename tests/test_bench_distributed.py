"""bench.py's OWN N > 1 path (one process per rank, instance sharding, the single all-gather of the result rows) executed end
to end on CPU: world sizes 2, 4 and 8 over gloo with the CPU emulation build of the kernels standing in for the GPU library.  The
gathered payload must equal what a single process computes for the same instance ids (bitwise: instances are independent
and the engine is deterministic)."""
import json
import os
import subprocess
import sys

import numpy as np

import scpp_amd
from conftest import ROOT


import pytest


@pytest.mark.parametrize("world,K,B,steps,warm", [(2, 8, 3, 2, 1), (4, 6, 2, 3, 0), (8, 5, 2, 1, 0)])
def test_bench_world_n_gloo_matches_single_process(model, emu_lib, tmp_path, world, K, B, steps, warm):
    """world 2, 4 and 8 (the first 8-GPU run must not be the first time this code sees N > 2), odd step counts"""
    maxit, seed = 4, 20260927
    dump = str(tmp_path / "rows.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29731 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warm),
           "--backend", "gloo", "--library", emu_lib, "--K", str(K), "--batch", str(B), "--max-iterations", str(maxit),
           "--no-cpu-baseline", "--no-extras", "--dump", dump,
           "--gather-chunk-mb", str(2 * (K * 18 + 10) * 8e-6 + 1e-7)]  # two rows per rank and collective
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["steps"] == steps and line["warmup"] == warm
    g = line["config"]["gather"]
    assert g["collectives"] == (steps * B + 1) // 2 and g["rows_per_collective"] == 2 and g["gathered_equals_local_bitwise"] is True
    assert g["bytes_per_rank_per_collective"] <= 2 * (K * 18 + 10) * 8
    r_ = line["roofline"]
    assert r_["kernel_time_s"] <= r_["timed_region_s"] * 1.001  # union of the launch spans, not their sum
    assert line["config"]["instances_timed"] == world * steps * B
    rows = np.load(dump)
    assert rows.shape == (world * steps * B, K * 18 + 10)
    got = scpp_amd.Context.unpack_stream_rows(rows, K)
    # the same instance ids through the plain batch entry point of one process
    alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=B, library=emu_lib, max_iterations=maxit).initialize()
    at = 0
    for rank in range(world):
        for i in range(steps):
            first = ((warm + i) * world + rank) * B
            x0 = model.randomized_initial_states(B, seed=seed, first=first)
            alg.solve(x0)
            ref = alg.getSolution()
            sl = slice(at, at + B)
            for key in ("X", "U", "sigma", "nu_norm", "sc_iters", "solves", "converged", "status", "ipm_iters"):
                assert np.array_equal(got[key][sl], ref[key]), (rank, i, key)
            at += B
    conv = int(got["converged"].sum())
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * steps - conv) < 1e-6 * max(conv, 1) + 1e-9
    # full payload of SURVEY 8(e): X, U, sigma, ||nu||_1, iterations, status all travel in the row
    assert (got["instance"].reshape(world, steps * B) == np.arange(steps * B)[None, :]).all()


def test_bench_gpus_flag_launches_the_ranks_itself(model, emu_lib, tmp_path):
    """plain `python bench.py --gpus 2` -- NO launcher around it, the form the driver uses -- must start two ranks by
    itself: n_gpus == 2, twice the instances, and the gathered rows bitwise equal to what one process computes for the same instance ids."""
    world, K, B, steps, warm, maxit, seed = 2, 6, 2, 3, 1, 3, 20260927
    dump = str(tmp_path / "rows.npy")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warm), "--backend", "gloo",
           "--library", emu_lib, "--K", str(K), "--batch", str(B), "--max-iterations", str(maxit), "--no-cpu-baseline", "--no-extras", "--dump", dump]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 prints, once
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == steps and line["warmup"] == warm and line["scaling"] == "weak"
    assert line["config"]["instances_timed"] == world * steps * B
    assert line["config"]["gather"]["gathered_equals_local_bitwise"] is True
    rows = np.load(dump)
    assert rows.shape == (world * steps * B, K * 18 + 10)
    got = scpp_amd.Context.unpack_stream_rows(rows, K)
    alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=B, library=emu_lib, max_iterations=maxit).initialize()
    at = 0
    for rank in range(world):
        for i in range(steps):
            x0 = model.randomized_initial_states(B, seed=seed, first=((warm + i) * world + rank) * B)
            alg.solve(x0)
            ref = alg.getSolution()
            for key in ("X", "U", "sigma", "nu_norm", "sc_iters", "solves", "converged", "status", "ipm_iters"):
                assert np.array_equal(got[key][at:at + B], ref[key]), (rank, i, key)
            at += B


def test_bench_gpus_flag_refuses_more_ranks_than_gpus(emu_lib):
    """--gpus N on a node with fewer than N GPUs (here: none) fails loudly before anything starts, and so does a launcher whose world size
    disagrees with --gpus: a mislabelled line would be worse than no line."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    import torch

    if torch.cuda.device_count() < 64:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode != 0 and "--gpus 64" in r.stderr and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo",
                        "--library", emu_lib], capture_output=True, text=True, timeout=600, env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_force_gather_single_rank(model, emu_lib, tmp_path):
    """--force-gather: ONE process creates a world-1 process group and runs the whole multi-GPU result path (view of the
    library's rows -> staging tensor -> chunked all_gather_into_tensor) inside the timed region; the dumped gathered rows are,
    bitwise, the rows of the plain single-process run.  (On the GPU box the same flag executes the RCCL branch:
    tests/test_gpu_parity.py::test_bench_force_gather_runs_rccl_on_one_gpu.)"""
    K, B, steps, maxit = 8, 3, 2, 4
    outs = []
    for extra, name in ((["--force-gather", "--gather-chunk-mb", "0.0025"], "g.npy"), ([], "p.npy")):
        dump = str(tmp_path / name)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "0", "--backend", "gloo",
               "--library", emu_lib, "--K", str(K), "--batch", str(B), "--max-iterations", str(maxit), "--no-cpu-baseline",
               "--no-extras", "--dump", dump] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        outs.append((line, np.load(dump)))
    (lg, rg), (lp, rp) = outs
    assert lg["config"]["gather"]["collectives"] == 3 and lg["config"]["gather"]["gathered_equals_local_bitwise"] is True
    assert lp["config"]["gather"] is None and lp["n_gpus"] == lg["n_gpus"] == 1
    assert rg.shape == rp.shape == (steps * B, K * 18 + 10)
    assert np.array_equal(rg.view(np.uint64), rp.view(np.uint64))


def test_bench_line_contract(emu_lib):
    """The ONE JSON line the driver parses: every key of the bench contract with the right type, the metric of BASELINE.json, the roofline and
    cpu_baseline objects, weak scaling, value = converged / time."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--library", emu_lib,
           "--K", "6", "--batch", "2", "--max-iterations", "3", "--no-extras"]  # (cpu_baseline bounds itself: ~12 s per solver)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # exactly one JSON line
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"]
    for key, typ in (("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool),
                     ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), (key, type(d[key]))
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    r_ = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r_, key
    assert r_["bound"] in ("hbm", "mfma") and r_["unit"] in ("GB/s", "TFLOP/s") and abs(r_["frac"] - r_["achieved"] / r_["peak"]) <= 1e-12
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - d["config"]["converged_fraction"] * d["config"]["instances_timed"]) <= 1e-6
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, (key, c)
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
