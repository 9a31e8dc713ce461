"""The THIRD model (round 6): Lander3dof, a point-mass powered-descent vehicle that the reference does not have, added through the
model-plugin path -- csrc/model_lander3dof.h (flow map), csrc/constraint_table.h: Lander3dofSC, csrc/sc_kernels.h: Lander3dofPlugin
(set-up + redimensionalisation, registered in `Plugins`), three entry points in the C ABI.  Nothing in the discretisation, the
simulation, the interior-point solver, the SC / SCvx loops or the streaming engine names it.

Checker: the oracle's restatement of the model in the reference's own plugin shape (oracle/models.hpp: Lander3dof) under the oracle's
LITERAL solver (the Epigraph-shaped problem of SCProblem.cpp / SCvxProblem.cpp solved by the generic sparse interior-point code, not the
structured twin, which exists for RocketQuat only).  CPU: the wave emulator of the kernel sources; GPU: the product library."""
import numpy as np
import pytest

import scpp_amd


def _model():
    return scpp_amd.Lander3dof().loadParameters()


def test_lander3dof_flow_map_matches_the_oracle(oracle):
    """systemFlowMap of the plugin (through scpp_hip_simulate's kernels is tested below); here: the generated analytic Jacobian rows vs the
    oracle's forward-mode AD of its own restatement, at seeded random points (tests/test_model_jacobian_rows.py checks the same header against
    AD of the C++ plugin itself)."""
    m = _model()
    rng = np.random.default_rng(5)
    par = m.flow_params()
    for _ in range(8):
        x = m.p.x_init * (1.0 + 0.1 * rng.standard_normal(7))
        u = np.array([1e4, -2e4, 3e5]) * (1.0 + 0.2 * rng.standard_normal(3))
        f, A, B = oracle.flow(oracle.LANDER3DOF, x, u, par)
        assert abs(f[0] + par[0] * np.linalg.norm(u)) <= 1e-12 * abs(f[0])
        assert np.allclose(f[1:4], x[4:7]) and np.allclose(f[4:7], u / x[0] + par[1:4])
        assert np.allclose(A[4:7, 0], -u / x[0] ** 2) and np.allclose(B[4:7], np.eye(3) / x[0]) and np.allclose(B[0], -par[0] * u / np.linalg.norm(u))


def _discretize_case(oracle, lib, tol):
    m = _model()
    K = 12
    rng = np.random.default_rng(11)
    X = np.stack([(1 - k / K) * m.p.x_init + k / K * m.p.x_final for k in range(K)]) * (1.0 + 0.01 * rng.standard_normal((K, 7)))
    U = np.tile([0.0, 0.0, 3.1e5], (K, 1)) + 2e4 * rng.standard_normal((K, 3))
    par, t = m.flow_params(), 11.0
    for mode, foh, vt in ((scpp_amd.MODE_FOH | scpp_amd.MODE_VT, True, True), (scpp_amd.MODE_FOH, True, False), (0, False, False)):
        ctx = scpp_amd.Context(scpp_amd.MODEL_LANDER3DOF, K, 2, library=lib)
        ctx.set_flow_params(np.tile(par, (2, 1)))
        Uin = U if foh else U[:K - 1]
        ctx.upload_traj(np.tile(X, (2, 1, 1)), np.tile(Uin, (2, 1, 1)), np.full(2, t))
        ctx.discretize(mode)
        got = ctx.download_dd()
        ref = oracle.discretize(oracle.LANDER3DOF, par, X, Uin, t, foh=foh, vt=vt)
        for g, r, used in zip(got, ref, (True, True, foh, vt, True)):
            if used:
                assert np.abs(g[1] - r).max() <= tol * max(1.0, np.abs(r).max())
        x1 = ctx.simulate(0.4, U[:2], U[1:3], X[:2])
        for b in range(2):
            assert np.abs(x1[b] - oracle.simulate(oracle.LANDER3DOF, par, 0.4, U[b], U[b + 1], X[b])).max() <= tol * np.abs(X[b]).max()
        ctx.close()


def test_emu_lander3dof_discretize_and_simulate_match_the_oracle(oracle, emu_lib):
    _discretize_case(oracle, emu_lib, 1e-10)


def _sc_case(oracle, lib, K, B):
    m = _model()
    alg = scpp_amd.SCAlgorithm(m, K=K, batch_max=B, library=lib).initialize()
    x0 = np.tile(m.x_init, (B, 1))
    x0[1:] = m.randomized_initial_states(B - 1, first=1)
    alg.solve(x0)
    out = alg.getSolution()
    assert (out["status"] == 0).all()
    worst = 0.0
    for b in range(min(B, 3)):
        sc = oracle.SC(oracle.LANDER3DOF, K=K); sc.set_x_init(x0[b]); sc.solve()
        X, U, t = sc.solution()
        assert out["sc_iters"][b] == sc.meta()["iterations"] and out["converged"][b] == sc.meta()["converged"]
        assert abs(out["sigma"][b] - t) <= 1e-5 * t  # north_star's 1e-5 (K = 30: 1.1e-6 after 15 iterations of a loop that stalls at ||nu||_1 = 0.065, like RocketQuat's)
        dX = np.abs(out["X"][b] - X).max() / np.abs(X).max()
        dU = np.abs(out["U"][b] - U).max() / np.abs(U).max()
        assert dX <= 1e-5 and dU <= 1e-4, (b, dX, dU)
        worst = max(worst, dX)
        assert (out["X"][b][:, 0] >= m.p.x_final[0] * (1 - 1e-9)).all()  # above the dry mass
    alg.ctx.close()
    return out, worst


def test_emu_lander3dof_sc_matches_the_oracle_literal_run(oracle, emu_lib):
    """SCAlgorithm on the emulator of the kernel sources: the structured solver instantiated for the Lander3dofSC table against the literal
    reference-shaped run (same SC iteration count, same convergence flag, trajectories to 1e-5)."""
    _sc_case(oracle, emu_lib, 10, 2)


def _scvx_case(oracle, lib, K, B, engine=None):
    m = _model()
    alg = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=B, library=lib).initialize()
    if engine is not None:
        alg.ctx.set_stream_engine(engine)
    x0 = np.tile(m.x_init, (B, 1))
    x0[1:] = m.randomized_initial_states(B - 1, first=1)
    alg.solve(x0)
    out = alg.getSolution()
    assert (out["status"] == 0).all()
    for b in range(min(B, 2)):
        sv = oracle.SCvx(K=K, model=oracle.LANDER3DOF); sv.set_x_init(x0[b]); sv.solve()
        X, U, t = sv.iterate(-1)
        assert out["sc_iters"][b] == sv.meta()["iterations"] and out["converged"][b] == sv.meta()["converged"]
        dX = np.abs(out["X"][b] - X).max() / np.abs(X).max()
        assert dX <= 1e-5, (b, dX)
    alg.ctx.close()
    return out


def test_emu_lander3dof_scvx_matches_the_oracle_literal_run(oracle, emu_lib):
    _scvx_case(oracle, emu_lib, 8, 2)


def _engines_case(lib, K, N, slots):
    """Streaming job through fewer slots than instances on BOTH engines (persistent kernel: its cost step gives two lanes one segment, and this
    model has an ODD number of states -- half 1 carries one padding slot; pool engine: one lane per segment) and the batch entry point:
    every instance's row is bitwise the same from all three."""
    m = _model()
    x0 = m.randomized_initial_states(N, first=10)
    a = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=N, library=lib).initialize()
    a.solve(x0)
    ref = a.getSolution()
    a.ctx.close()
    engines = {}
    for engine in (scpp_amd._lib.STREAM_PERSISTENT, scpp_amd._lib.STREAM_POOLS):
        s = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=slots, library=lib).initialize()
        s.ctx.set_stream_engine(engine)
        n = s.solveStream(x0, slots=slots)
        rows = s.getStreamSolution()
        engines[engine] = s.ctx.stream_rounds()["pools"]
        for k in ("X", "U", "sc_iters", "solves", "ipm_iters", "status"):
            assert np.array_equal(rows[k], ref[k]), (engine, k)
        s.ctx.close()
    assert engines[scpp_amd._lib.STREAM_PERSISTENT] == 0 and engines[scpp_amd._lib.STREAM_POOLS] >= 1  # (pools = 0: the persistent kernel ran the job)
    return n, ref


def test_emu_lander3dof_stream_rows_equal_the_batch_solve_on_both_engines(emu_lib):
    _engines_case(emu_lib, 8, 5, 2)


@pytest.mark.gpu
def test_lander3dof_discretize_and_simulate_match_the_oracle_on_gpu(oracle, hip_lib):
    _discretize_case(oracle, hip_lib, 1e-10)


@pytest.mark.gpu
def test_lander3dof_sc_and_scvx_match_the_oracle_on_gpu(oracle, hip_lib):
    """The third model on the product library: SC_oneshot and SCvx at the shipped K = 30, 64 randomised instances; three / two of them
    against the oracle's literal run (iteration counts, convergence flags, trajectories to 1e-5), all of them solved without a solver
    failure; a streaming job through fewer slots than instances returns the batch entry point's rows bitwise."""
    out, worst = _sc_case(oracle, hip_lib, 30, 64)
    vx = _scvx_case(oracle, hip_lib, 30, 64)
    n, ref = _engines_case(hip_lib, 30, 64, 24)
    print("Lander3dof on the GPU: SC converged %d / 64 (worst rel dX vs the literal run %.1e), SCvx converged %d / 64 in %.1f iterations on average; "
          "streaming job of 64 instances through 24 slots: %d converged, rows bitwise the batch solve's on the persistent kernel and on the pool engine"
          % (int(out["converged"].sum()), worst, int(vx["converged"].sum()), float(vx["sc_iters"].mean()), n))
