"""N>1 path on CPU: world_size=2 over gloo, the emulation build standing in for the GPU library.
Checks the sharding + all-gather of results against a single-process solve of the same instances."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import scpp_amd
from scpp_amd.distributed import solve_sharded, mpc_solve_sharded, shard_range
emu, total, K, outdir = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
model = scpp_amd.RocketQuat().loadParameters()
lo, hi = shard_range(total, dist.get_world_size(), dist.get_rank())
mode = sys.argv[6] if len(sys.argv) > 6 else "sc"
if mode == "mpc":
    m2 = scpp_amd.Rocket2D().loadParameters(); m2.p.constrain_initial_final = False
    alg = scpp_amd.MPCAlgorithm(m2, batch_max=hi - lo, library=emu).initialize()
    res = mpc_solve_sharded(alg, m2, total, 20260927, dist=dist)
    np.savez(os.path.join(outdir, f"rank{dist.get_rank()}.npz"), **res)
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
if mode == "scvx":
    alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=hi - lo, library=emu, max_iterations=4).initialize()
else:
    alg = scpp_amd.SCAlgorithm(model, K=K, batch_max=hi - lo, library=emu).initialize()
res = solve_sharded(alg, model, total, 20260927, dist=dist)
np.savez(os.path.join(outdir, f"rank{dist.get_rank()}.npz"), **res)
dist.barrier(); dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("mode,port", [("sc", "29541"), ("scvx", "29542"), ("mpc", "29543")])
def test_world2_gloo_sharded_solve_matches_single_process(emu_lib, model, tmp_path, mode, port):
    import scpp_amd
    from scpp_amd.distributed import shard_range, solve_sharded

    total, K = 5, 6  # uneven shards: 3 + 2
    assert shard_range(total, 2, 0) == (0, 3) and shard_range(total, 2, 1) == (3, 5)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, emu_lib, str(total), str(K), str(tmp_path), mode], env=e))
    for p in procs:
        assert p.wait(timeout=600) == 0
    if mode == "mpc":
        from scpp_amd.distributed import mpc_solve_sharded
        m2 = scpp_amd.Rocket2D().loadParameters(); m2.p.constrain_initial_final = False
        alg = scpp_amd.MPCAlgorithm(m2, batch_max=total, library=emu_lib).initialize()
        single = mpc_solve_sharded(alg, m2, total, 20260927)
        for r in range(2):
            got = np.load(tmp_path / f"rank{r}.npz")
            for key in ("u0", "cost", "status", "iters"):
                assert np.array_equal(got[key], single[key]), (r, key)
        assert (single["status"] == 0).all()
        return
    if mode == "scvx":
        alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=total, library=emu_lib, max_iterations=4).initialize()
    else:
        alg = scpp_amd.SCAlgorithm(model, K=K, batch_max=total, library=emu_lib).initialize()
    single = solve_sharded(alg, model, total, 20260927)
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        for key in ("X", "U", "sigma", "nu_norm", "sc_iters", "converged"):
            assert np.array_equal(got[key], single[key]), (r, key)
