"""Instances of a batch are independent problems ("own x_init, own scaling, own SC loop; no cross-instance data"): the row
an entry point returns for an instance may depend neither on the instance's position in the batch, nor on its neighbours, nor on what
the context solved before.  Checked bitwise on the CPU wave emulator for the three batch entry points (SCAlgorithm, SCvxAlgorithm,
MPCAlgorithm) by solving a batch, then -- on the SAME context -- a permuted, shorter batch (stale rows of the longer one behind it) and a
batch of copies of one instance.  (The streaming engine's version of this property is tests/test_emu_stream_fuzz.py.)"""
import numpy as np
import pytest

import scpp_amd


def _rows_equal(a, b, keys, what):
    for key in keys:
        assert np.array_equal(a[key], b[key]), (what, key)


def _take(o, idx, keys):
    return {k: o[k][idx] for k in keys}


def test_emu_sc_rows_do_not_depend_on_batch_composition(model, emu_lib):
    _sc_case(model, emu_lib, 7, 4)


def test_emu_scvx_rows_do_not_depend_on_batch_composition(model, emu_lib):
    _scvx_case(model, emu_lib, 7, 5)


def test_emu_mpc_rows_do_not_depend_on_batch_composition(emu_lib):
    _mpc_case(emu_lib)


@pytest.mark.gpu
def test_rows_do_not_depend_on_batch_composition_on_gpu(model, hip_lib):
    """the same three properties on hardware, SC / SCvx at the bench's K = 50"""
    _sc_case(model, hip_lib, 50, 15)
    _scvx_case(model, hip_lib, 50, 12)
    _mpc_case(hip_lib)


def _sc_case(model, emu_lib, K, maxit):
    keys = ("X", "U", "sigma", "nu_norm", "sum_delta", "sc_iters", "ipm_iters", "status", "converged")
    B = 6
    alg = scpp_amd.SCAlgorithm(model, K=K, batch_max=B, library=emu_lib).initialize()
    alg.opts.max_iterations = maxit
    x0 = model.randomized_initial_states(B, first=4100)
    alg.solve(x0)
    full = alg.getSolution()
    assert (full["status"] == 0).all()
    perm = np.array([4, 0, 5, 2])
    alg.solve(x0[perm])  # shorter batch on the same context: rows 4, 5 of the first solve lie behind it
    _rows_equal(alg.getSolution(), _take(full, perm, keys), keys, "SC permuted subset")
    alg.solve(np.repeat(x0[3:4], 5, axis=0))
    rep = alg.getSolution()
    for j in range(5):
        _rows_equal(_take(rep, j, keys), _take(full, 3, keys), keys, "SC copies of one instance")
    alg.ctx.close()


def _scvx_case(model, emu_lib, K, maxit):
    keys = ("X", "U", "sigma", "nu_norm", "nonlinear_cost", "trust_region", "sc_iters", "solves", "ipm_iters", "status", "converged")
    B = 6
    alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=B, library=emu_lib, max_iterations=maxit).initialize()
    x0 = model.randomized_initial_states(B, first=5200)
    alg.solve(x0)
    full = alg.getSolution()
    assert (full["status"] == 0).all()
    perm = np.array([5, 1, 3])
    alg.solve(x0[perm])
    _rows_equal(alg.getSolution(), _take(full, perm, keys), keys, "SCvx permuted subset")
    # the same context again with the full batch reversed (every slot now holds another instance's warm state from two solves ago)
    alg.solve(x0[::-1].copy())
    _rows_equal(alg.getSolution(), _take(full, np.arange(B)[::-1], keys), keys, "SCvx reversed")
    alg.ctx.close()


def _mpc_case(emu_lib):
    keys = ("X", "U", "iters", "status", "cost")
    m2 = scpp_amd.Rocket2D().loadParameters()
    m2.p.constrain_initial_final = False  # model.info: "enable for SC and disable for MPC/LQR"
    alg = scpp_amd.MPCAlgorithm(m2, batch_max=8, library=emu_lib).initialize()
    x0 = m2.randomized_initial_states(8, first=6300)
    x0[2, 4] = 1.3  # tilt outside its box: infeasible at k = 0 -- a failed neighbour must not disturb the others
    alg.setInitialState(x0); alg.setFinalState(m2.p.x_final)
    alg.solve()
    full = alg.getSolution()
    assert full["status"][2] == -3 and (np.delete(full["status"], 2) == 0).all()
    ok = np.array([7, 0, 4, 1, 6])
    alg.setInitialState(x0[ok]); alg.setFinalState(m2.p.x_final)
    assert alg.solve() == 5
    _rows_equal(alg.getSolution(), _take(full, ok, keys), keys, "MPC permuted subset without the infeasible neighbour")
    alg.ctx.close()
