"""The WHOLE SCAlgorithm loop on the shipped RocketQuat scenario against a solver the build did not write.

tests/golden/rocketquat_sc_loop_K5.npz holds ||nu||_1, sum(delta), sigma per SC iteration from scipy `trust-constr` driving
the loop of SCAlgorithm::solve (SCAlgorithm.cpp:66-189) on an NLP restatement that shares no code with the oracle, the kernels or
scpp_amd (tests/golden/generate_sc_loop_goldens.py).  What they pin:
  * the iteration does NOT meet the convergence test (sum(delta) < 1e-3 and ||nu||_1 < 1e-5, SCAlgorithm.cpp:131): it stalls at a
    fixed point -- the iterate stops moving (sum(delta) -> 1e-10) while the virtual control stays at ||nu||_1 = 6.9e-4 (K = 5, after
    one doubling of the trust-region weight in iteration 2, where ||nu||_1 dips below nu_tol) / 0.0193 (K = 15, no doubling) --
    which is why `converged_fraction` is 0 in SC mode;
  * both oracle solvers (literal ECOS-style, structured twin) and the device path follow the same sequence iteration by iteration.
Tolerances: see _check (tight for the first three iterations, where trust-constr converges; 15 % on ||nu||_1 afterwards, where it
stops short at the non-smooth trust-region centres -- with a HIGHER objective than the solvers under test).
"""
import os

import numpy as np
import pytest

import scpp_amd
from conftest import GOLDEN


def _golden(K):
    f = os.path.join(GOLDEN, "rocketquat_sc_loop_K%d.npz" % K)
    if not os.path.exists(f):
        pytest.skip("golden %s not generated" % os.path.basename(f))
    return np.load(f)


NU_TOL, DELTA_TOL = 1e-5, 1e-3  # SC.info
# K = 5 only.  A K = 15 record of the scipy loop was planned in round 2 and never generated (trust-constr needs hours there), so its tests were
# skipped for three rounds: removed in round 5.  The SC loop at K = 15 is pinned differently and more sharply since round 4 -- every sub-problem
# along the device's own path against Kelley cutting planes over HiGHS (tests/test_independent_path_audit.py: _sc_path_audit, CPU emulator
# and, since round 5, the GPU).
KS = [5]


def _check(g, nu, sd, sigma, what):
    """Iteration by iteration against the scipy record.  Iterations 1-3 start from smooth points and trust-constr reaches the
    sub-problem optimum to ~1e-7 in the objective: ||nu||_1 within 5e-3 (of max(||nu||_1, 1e-3)), sigma within 1e-4.  From iteration 4 on the linearisation
    point sits at the apex of the 50 trust-region cones of the previous solution (||x - xbar|| = 0: not differentiable), where
    trust-constr stops 0.1 .. 1 % short of the optimum our solvers find (its objective is HIGHER than theirs, never lower): there
    the comparison is ||nu||_1 within 15 %, sigma within 5e-4, and sum(delta) on the same side of delta_tol.  What the record pins
    at every iteration regardless: the weight-doubling decisions (||nu||_1 vs nu_tol) and the verdict (no convergence)."""
    n = int(g["iterations"])
    assert len(nu) >= n, what
    for it in range(n):
        tight = it < 3
        assert (nu[it] < NU_TOL) == (g["norm1_nu"][it] < NU_TOL), (what, it, nu[it], g["norm1_nu"][it])
        assert abs(nu[it] - g["norm1_nu"][it]) <= (5e-3 if tight else 0.15) * max(g["norm1_nu"][it], 1e-3), (what, it, nu[it], g["norm1_nu"][it])
        assert abs(sigma[it] - g["sigma"][it]) <= (1e-4 if tight else 5e-4) * g["sigma"][it], (what, it, sigma[it], g["sigma"][it])
        if tight:
            assert abs(sd[it] - g["sum_delta"][it]) <= 5e-3 * max(g["sum_delta"][it], 1e-2), (what, it, sd[it], g["sum_delta"][it])
        else:
            assert (sd[it] < DELTA_TOL) == (g["sum_delta"][it] < DELTA_TOL) or abs(sd[it] - g["sum_delta"][it]) <= 0.1 * g["sum_delta"][it], (what, it)


@pytest.mark.parametrize("K", KS)
def test_scipy_loop_stalls_without_converging(K):
    g = _golden(K)
    n = int(g["iterations"])
    assert n >= 3 and int(g["converged"]) == 0
    assert (g["constr_violation"] < 1e-5).all()
    # the recorded weights follow SCAlgorithm.cpp:112-115 from the recorded norms
    w = 50.0
    for it in range(n):
        assert g["weight_trx"][it] == w
        if g["norm1_nu"][it] < NU_TOL:
            w *= 2.0
    if n >= 6:  # the stall: the iterate stops moving (sum(delta) -> 0) while the virtual control stays far above nu_tol
        assert g["sum_delta"][n - 1] < 1e-3 and g["norm1_nu"][n - 1] > 50 * NU_TOL
        assert abs(g["norm1_nu"][n - 1] - g["norm1_nu"][n - 2]) < 1e-3 * g["norm1_nu"][n - 1]


@pytest.mark.parametrize("K", KS)
@pytest.mark.parametrize("solver", [0, 1])
def test_oracle_solvers_follow_the_scipy_loop(oracle, K, solver):
    g = _golden(K)
    sc = oracle.SC(oracle.ROCKETQUAT, K=K); sc.set_solver(solver)
    assert sc.solve() == 0
    inf = sc.info()
    assert sc.meta()["converged"] == 0 and sc.meta()["iterations"] == 15
    _check(g, inf[:, 0], inf[:, 1], inf[:, 3], "oracle solver %d" % solver)


def _device_loop(model, K, lib):
    alg = scpp_amd.SCAlgorithm(model, K=K, batch_max=1, library=lib).initialize()
    alg.ctx.sc_setup(model.p, alg.opts, model.x_init[None])
    nu, sd, sg = [], [], []
    for _ in range(int(alg.opts.max_iterations)):
        alg.ctx.sc_iterate()
        o = alg.ctx.download()
        nu.append(float(o["nu_norm"][0])); sd.append(float(o["sum_delta"][0])); sg.append(float(o["sigma"][0]))
    conv = int(alg.ctx.download()["converged"][0])
    alg.ctx.close()
    return nu, sd, sg, conv


@pytest.mark.parametrize("K", KS)
def test_device_path_follows_the_scipy_loop_emulated(model, emu_lib, K):
    g = _golden(K)
    nu, sd, sg, conv = _device_loop(model, K, emu_lib)
    assert conv == 0
    _check(g, nu, sd, sg, "device (emulated)")


@pytest.mark.gpu
@pytest.mark.parametrize("K", KS)
def test_device_path_follows_the_scipy_loop_on_gpu(model, hip_lib, K):
    g = _golden(K)
    nu, sd, sg, conv = _device_loop(model, K, hip_lib)
    assert conv == 0
    _check(g, nu, sd, sg, "device (HIP)")
    print("SC loop K=%d vs scipy trust-constr over %d iterations: final ||nu||_1 %.6f (scipy %.6f), sigma %.6f (scipy %.6f)"
          % (K, int(g["iterations"]), nu[int(g["iterations"]) - 1], g["norm1_nu"][-1], sg[int(g["iterations"]) - 1], g["sigma"][-1]))
