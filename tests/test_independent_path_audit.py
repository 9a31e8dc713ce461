"""Sub-problems ALONG the device's own SCvx path against an optimum that owes nothing to this repository's solvers.

tests/test_subproblem_pin.py pins the FIRST sub-problem (linearised at the initial guess).  Here the nominal RocketQuat run of the device
(CPU wave emulator) is followed for a few iterations; every accepted sub-problem -- linearised at the device's previous iterate, with the
trust radius the accept / reject history had produced by then (SCvxAlgorithm.cpp:118-152) -- is restated by the independent numpy / sympy /
DOP853 code of tests/golden/generate_subproblem_goldens.py (no line of oracle/, scpp_amd or the kernels) and solved by Kelley's cutting
planes over HiGHS (generate_subproblem_cut_goldens.py).  The device's candidate must be a point of that problem (equalities 1e-9, cones and
rows -1e-9) whose objective w_vc ||nu||_1 -- nu taken as the candidate's defect in the RESTATED linearised dynamics -- is within 5e-5 of the
LP-based optimum and not below it by more than 1e-6 (the bars of the literal audit, tests/test_gpu_parity.py: BAR_CERT_*).
What this adds to the literal audit (tests/scvx_audit.py): the checker there is the oracle's own literal solver and discretisation.
Checked the same way: the nonlinear cost J the device reports for every iterate (SCvxAlgorithm.cpp:262-278: sum over the segments of
||x_propagated - x_{k+1}||_1) against DOP853 propagation of the restated flow map, and -- on the iterations whose candidate was accepted at
the first attempt, where the previous J is known -- the device's rho = dJ / dL and its trust-radius update (SCvxAlgorithm.cpp:118-152)."""
import os
import sys

import numpy as np
import pytest

import scpp_amd
from conftest import GOLDEN

sys.path.insert(0, GOLDEN)

K = 15
N_IT = 6
W_VC = 1000.0


def _nondim(X, U, ms, rs):
    X = X.copy(); U = U.copy()
    X[:, 0] /= ms; X[:, 1:7] /= rs; U[:, :3] /= ms * rs; U[:, 3] /= ms * rs * rs
    return X, U


def _nonlinear_cost(G, sc, X, U, t, K=K):
    """SCvxAlgorithm.cpp:262-278 + simulation.cpp:25-42 with the restated flow map, first-order-hold inputs, DOP853 instead of RKF78"""
    import sympy as sp
    from scipy.integrate import solve_ivp
    from generate_goldens import rocketquat_sym

    x_, u_, p_, f_ = rocketquat_sym()
    fn = sp.lambdify([x_, u_, p_], f_, "numpy")
    dt, J = t / (K - 1), 0.0
    for k in range(K - 1):
        rhs = lambda tau, x: np.asarray(fn(x, U[k] + tau / dt * (U[k + 1] - U[k]), sc["par"]), dtype=float).ravel()  # noqa: E731
        xe = solve_ivp(rhs, [0, dt], X[k], method="DOP853", rtol=1e-13, atol=1e-16).y[:, -1]
        J += np.abs(xe - X[k + 1]).sum()
    return J


def _device_path(alg, x0, n):
    """iterate j of the nominal run = the run capped at max_iterations = j (deterministic), with what the device reports about it"""
    path, keep = [], alg._max_iterations
    try:
        for j in range(n + 1):
            alg._max_iterations = j
            alg.solve(x0)
            o = alg.getSolution()
            assert o["status"][0] == 0
            path.append({k: np.array(o[k][0]).copy() for k in ("X", "U", "trust_region", "solves", "sc_iters", "converged", "nonlinear_cost", "nu_norm", "last_decision")})
    finally:
        alg._max_iterations = keep
    return path


def test_emu_scvx_path_sub_problems_against_independent_cutting_planes(emu_lib):
    _path_audit(emu_lib, N_IT)


@pytest.mark.gpu
def test_scvx_path_sub_problems_against_independent_cutting_planes_on_gpu(hip_lib):
    _path_audit(hip_lib, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("instance", [None, 3, 11])
def test_scvx_path_at_the_size_of_the_metric_against_independent_cutting_planes_on_gpu(hip_lib, instance):
    """the independent audit at the size the metric is quoted on -- K = 50 -- on the GPU, for the nominal instance and two of
    the bench's randomised instances (their own mass / distance scaling): five accepted sub-problems each, rejected candidates on the way
    included, the nonlinear cost, rho and the radius rule recomputed independently.  Solved at test time (a cutting-plane run over HiGHS per
    sub-problem): the device path is deterministic per BUILD only -- a recompilation moves it at rounding level, which golden optima keyed
    on device iterates would not survive."""
    _path_audit(hip_lib, 5, K=50, instance=instance, workers=5)


def _audit_one(task):
    """One accepted sub-problem of a device path, checked with code that shares nothing with the repository's solvers (pure CPU; runs in a worker
    process when several are audited at once).  Returns (gap, J, L) after asserting feasibility and the objective bars."""
    import generate_subproblem_cut_goldens as C
    import generate_subproblem_goldens as G

    Kn, x_dim, j, Xb, Ub, r_used, Xc, Uc = task
    G.K = Kn
    sc = G.scenario(x_dim)
    dd = G.discretize(sc, Xb, Ub, sc["final_time"], False)
    pb = G.SubProblem(sc, Xb, Ub, sc["final_time"], dd, "scvx", dict(vc=W_VC, tr=r_used))
    _, info = C.solve_cuts(pb, verbose=False)
    assert max(info["cone_violation"], info["eq_violation"], info["lin_violation"]) <= 1e-9
    # the device's candidate as a point of the restated problem: nu := its defect in the restated dynamics
    A, B, Cm, S, Z = dd
    nu = np.array([Xc[k + 1] - (A[k] @ Xc[k] + B[k] @ Uc[k] + Cm[k] @ Uc[k + 1] + Z[k]) for k in range(Kn - 1)])
    v = np.concatenate([Xc.ravel(), Uc.ravel(), np.maximum(nu, 0).ravel(), np.maximum(-nu, 0).ravel()])
    assert np.abs(pb.eq(v)).max() <= 1e-9, ("equalities", j + 1, float(np.abs(pb.eq(v)).max()))
    assert pb.ineq(v).min() >= -1e-9, ("rows / cones", j + 1, float(pb.ineq(v).min()))
    gap = (pb.cost(v) - info["objective"]) / info["objective"]
    assert -1e-6 <= gap <= 5e-5, ("objective", j + 1, pb.cost(v), info["objective"])
    J = _nonlinear_cost(G, sc, Xc, Uc, sc["final_time"], Kn)
    return gap, J, pb.cost(v) / W_VC


def _path_audit(emu_lib, n_it, K=K, instance=None, workers=1):
    m = scpp_amd.RocketQuat().loadParameters()
    x_init = m.x_init if instance is None else m.randomized_initial_states(1, first=instance)[0]
    ms, rs = float(x_init[0]), float(np.linalg.norm(x_init[1:4]))  # rocketQuat.cpp:293-294
    alg = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=1, library=emu_lib).initialize()
    o_ = alg.opts
    alpha, beta, rho_1, rho_2 = float(o_.alpha), float(o_.beta), float(o_.rho_1), float(o_.rho_2)
    path = _device_path(alg, x_init[None], n_it)
    alg.ctx.close()
    Xb, Ub = _nondim(path[0]["X"], path[0]["U"], ms, rs)
    r_prev, solves_prev = float(path[0]["trust_region"]), 0
    tasks, meta = [], []
    rejected = 0
    for j, st in enumerate(path[1:]):
        n_rej = int(st["solves"] - solves_prev) - 1
        assert st["sc_iters"] == j + 1 and n_rej >= 0 and st["converged"] == 0
        rejected += n_rej
        r_used = r_prev / alpha ** n_rej
        Xc, Uc = _nondim(st["X"], st["U"], ms, rs)
        tasks.append((K, None if instance is None else x_init.copy(), j, Xb, Ub, r_used, Xc, Uc))
        meta.append((n_rej, r_used))
        Xb, Ub, r_prev, solves_prev = Xc, Uc, float(st["trust_region"]), int(st["solves"])
    if workers > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor

        with ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as ex:  # (spawn: the parent holds a HIP context)
            results = list(ex.map(_audit_one, tasks))
    else:
        results = [_audit_one(t) for t in tasks]
    gaps, rho_checked, J_prev = [], 0, None
    for j, (st, (n_rej, r_used), (gap, J, L)) in enumerate(zip(path[1:], meta, results)):
        gaps.append(gap)
        # the nonlinear cost of the candidate, and the accept / radius rule where the previous J is known (no rejection in between)
        assert abs(J - float(st["nonlinear_cost"])) <= 1e-8 * J, ("J", j + 1, J, float(st["nonlinear_cost"]))
        assert abs(L - float(st["nu_norm"])) <= 1e-6 * L, ("L", j + 1, L, float(st["nu_norm"]))  # (the solver's norm1_nu: tight to its 1e-8 tolerances)
        rho_dev, dJ_dev, dL_dev, code = [float(x) for x in st["last_decision"]]
        if j == 0:
            assert code == 2.0 and float(st["trust_region"]) == r_used  # first pass: J is stored, nothing is decided (SCvxAlgorithm.cpp:109-113)
        elif n_rej == 0:
            dJ, dL = J_prev - J, J_prev - L
            rho = dJ / dL
            assert code == 1.0 and abs(dJ - dJ_dev) <= 1e-7 * abs(dL) and abs(dL - dL_dev) <= 1e-7 * abs(dL) and abs(rho - rho_dev) <= 1e-6
            r_next = r_used / alpha if rho < rho_1 else (r_used * beta if rho >= rho_2 else r_used)
            assert min(abs(rho - rho_1), abs(rho - rho_2)) > 1e-4 and abs(float(st["trust_region"]) - r_next) <= 1e-12 * r_next, ("radius", j + 1, rho)
            rho_checked += 1
        J_prev = J
    assert rejected >= 1 and rho_checked >= 1  # the path exercises both: rejected candidates and first-attempt acceptances
    print("independent audit of %d sub-problems along the device path (%s instance, K = %d, %d rejected candidates on the way, rho and the radius rule "
          "checked on %d iterations): relative objective gaps %s" % (len(gaps), "nominal" if instance is None else "randomised #%d" % instance, K, rejected,
                                                                     rho_checked, ["%.1e" % g for g in gaps]))


# ---------------------------------------------------------------------------------------------------------------------
# The same for the SC mode (SCAlgorithm.cpp:66-132 + SCProblem.cpp:6-138: free final time, soft trust region).  Every sub-problem of the
# nominal run is linearised at the device's previous iterate (X, U, sigma) -- variable-time DOP853 sensitivities -- and solved independently;
# the device's next iterate must be a point of it whose objective w_t sigma + w_vc ||nu||_1 + w_trt (sigma - sigma_bar)^2 + w_trx sum ||(dx, du)||
# is the LP-based optimum to 1e-6.  The soft trust region makes this optimum unique in X and U: they are compared as well (1e-4: what an LP
# vertex of the outer approximation determines).
# ---------------------------------------------------------------------------------------------------------------------
def _sc_path_audit(lib, n_it):
    import generate_subproblem_cut_goldens as C
    import generate_subproblem_goldens as G
    import scvx_audit

    G.K = K
    sc = G.scenario()
    ms, rs = sc["m_scale"], sc["r_scale"]
    m = scpp_amd.RocketQuat().loadParameters()
    alg = scpp_amd.SCAlgorithm(m, K=K, batch_max=1, library=lib).initialize()
    o_ = alg.opts
    w = dict(t=float(o_.weight_time), trt=float(o_.weight_trust_region_time), trx=float(o_.weight_trust_region_trajectory), vc=float(o_.weight_virtual_control))
    assert (w["t"], w["trt"], w["trx"], w["vc"]) == (1.0, 1.0, 50.0, 1000.0)  # shipped RocketQuat SC.info
    Xb, Ub, sb = G.initial_trajectory(sc)
    gaps, dX, dU = [], [], []
    for j in range(1, n_it + 1):
        o = scvx_audit.sc_device_iterate(alg, m.x_init[None], j)
        assert o["status"][0] == 0 and o["sc_iters"][0] == j
        Xc, Uc = _nondim(o["X"][0], o["U"][0], ms, rs)
        sig = float(o["sigma"][0])
        dd = G.discretize(sc, Xb, Ub, sb, True)
        pb = G.SubProblem(sc, Xb, Ub, sb, dd, "sc", w)
        x_lp, info = C.solve_cuts(pb, verbose=False)
        # (the LP solver's own accuracy; rounds it could only finish with its default tolerances leave equality residuals of some 1e-9)
        assert info["cone_violation"] <= 1e-9 and max(info["eq_violation"], info["lin_violation"]) <= 1e-7
        A, B, Cm, S, Z = dd
        nu = np.array([Xc[k + 1] - (A[k] @ Xc[k] + B[k] @ Uc[k] + Cm[k] @ Uc[k + 1] + S[k] * sig + Z[k]) for k in range(K - 1)])
        delta = np.sqrt(((Xc - Xb) ** 2).sum(axis=1) + ((Uc - Ub) ** 2).sum(axis=1)) + 1e-12
        v = np.concatenate([Xc.ravel(), Uc.ravel(), np.maximum(nu, 0).ravel(), np.maximum(-nu, 0).ravel(), delta, [sig, (sig - sb) ** 2]])
        assert np.abs(pb.eq(v)).max() <= 1e-9, ("equalities", j, float(np.abs(pb.eq(v)).max()))
        assert pb.ineq(v).min() >= -1e-9, ("rows / cones", j, float(pb.ineq(v).min()))
        gap = (pb.cost(v) - info["objective"]) / info["objective"]
        assert -1e-6 <= gap <= 1e-6, ("objective", j, pb.cost(v), info["objective"])
        assert abs(float(o["sum_delta"][0]) - (delta.sum() - K * 1e-12)) <= 1e-6 * delta.sum() and abs(float(o["nu_norm"][0]) - np.abs(nu).sum()) <= 1e-6 * np.abs(nu).sum() + 1e-9
        Xl, Ul = pb.split(x_lp)[:2]
        gaps.append(gap); dX.append(float(np.abs(Xl - Xc).max() / np.abs(Xc).max())); dU.append(float(np.abs(Ul - Uc).max() / np.abs(Uc).max()))
        # (an LP vertex of the outer approximation with cones met to 1e-10 locates the optimum to ~1e-5 only: the objective is the sharp statement)
        assert abs(pb.split(x_lp)[5] - sig) <= 1e-4 * sig and dX[-1] <= 1e-4 and dU[-1] <= 1e-4, ("point", j, dX[-1], dU[-1])
        Xb, Ub, sb = Xc, Uc, sig
        if o["nu_norm"][0] < float(o_.nu_tol):  # SCAlgorithm.cpp:112-115: the next sub-problem weighs the trust region twice as much
            w = dict(w, trx=2.0 * w["trx"])
    alg.ctx.close()
    print("independent audit of %d SC sub-problems along the device path (K = %d): relative objective gaps %s, states within %.1e, inputs within %.1e of "
          "the LP-based optimum" % (len(gaps), K, ["%.1e" % g for g in gaps], max(dX), max(dU)))


def test_emu_sc_path_sub_problems_against_independent_cutting_planes(emu_lib):
    _sc_path_audit(emu_lib, 4)


# ---------------------------------------------------------------------------------------------------------------------
# The second model: Rocket2D SC (K = 25, the reference's SC.info) and SCvx (nondimensionalised; K = 15) along the device path, against the
# independent restatement of tests/golden/generate_rocket2d_cut_goldens.py.
# ---------------------------------------------------------------------------------------------------------------------
def _nondim2d(X, U, ms, rs):
    X = X.copy(); U = U.copy()
    X[:, :4] /= rs; U[:, 1] /= ms * rs
    return X, U


def _r2d_candidate(R, pb, dd, Kn, Xc, Uc, sig, Xb, Ub, sb):
    A, B, Cm, S, Z = dd
    nu = np.array([Xc[k + 1] - (A[k] @ Xc[k] + B[k] @ Uc[k] + Cm[k] @ Uc[k + 1] + S[k] * sig + Z[k]) for k in range(Kn - 1)])
    parts = [Xc.ravel(), Uc.ravel(), np.maximum(nu, 0).ravel(), np.maximum(-nu, 0).ravel()]
    if pb.mode == "sc":
        delta = np.sqrt(((Xc - Xb) ** 2).sum(axis=1) + ((Uc - Ub) ** 2).sum(axis=1)) + 1e-12
        parts += [delta, [sig, (sig - sb) ** 2]]
    v = np.concatenate(parts)
    assert np.abs(pb.eq(v)).max() <= 1e-9 and pb.ineq(v).min() >= -1e-9, (float(np.abs(pb.eq(v)).max()), float(pb.ineq(v).min()))
    return v, float(np.abs(nu).sum())


def test_emu_rocket2d_sc_path_sub_problems_against_independent_cutting_planes(emu_lib):
    import generate_rocket2d_cut_goldens as R
    import generate_subproblem_cut_goldens as C
    import scvx_audit

    Kn = 25
    sc = R.scenario(True)
    ms, rs = sc["m_scale"], sc["r_scale"]
    m = scpp_amd.Rocket2D().loadParameters()
    alg = scpp_amd.SCAlgorithm(m, K=Kn, batch_max=1, library=emu_lib).initialize()
    o_ = alg.opts
    w = dict(t=float(o_.weight_time), trt=float(o_.weight_trust_region_time), trx=float(o_.weight_trust_region_trajectory), vc=float(o_.weight_virtual_control))
    assert (w["t"], w["trt"], w["trx"], w["vc"]) == (1.0, 1.0, 1.0, 1000.0) and o_.nondimensionalize == 1  # Rocket2D/SC.info
    Xb, Ub, sb = R.initial_trajectory(sc, Kn)
    gaps, doubled = [], 0
    for j in range(1, 5):
        o = scvx_audit.sc_device_iterate(alg, m.x_init[None], j)
        assert o["status"][0] == 0 and o["sc_iters"][0] == j
        Xc, Uc = _nondim2d(o["X"][0], o["U"][0], ms, rs)
        sig = float(o["sigma"][0])
        dd = R.discretize(sc, Kn, Xb, Ub, sb, True)
        pb = R.SubProblem(sc, Kn, Xb, Ub, sb, dd, "sc", w)
        _, info = C.solve_cuts(pb, verbose=False)
        assert info["cone_violation"] <= 1e-9 and max(info["eq_violation"], info["lin_violation"]) <= 1e-7
        v, l1 = _r2d_candidate(R, pb, dd, Kn, Xc, Uc, sig, Xb, Ub, sb)
        gap = (pb.cost(v) - info["objective"]) / info["objective"]
        assert -1e-6 <= gap <= 1e-6 and abs(float(o["nu_norm"][0]) - l1) <= 1e-6 * l1 + 1e-9, ("Rocket2D SC", j, pb.cost(v), info["objective"])
        gaps.append(gap)
        Xb, Ub, sb = Xc, Uc, sig
        if o["nu_norm"][0] < float(o_.nu_tol):  # SCAlgorithm.cpp:112-115 (here from the second iteration on: the virtual control has vanished)
            w = dict(w, trx=2.0 * w["trx"]); doubled += 1
    alg.ctx.close()
    assert doubled >= 1  # the path exercises the weight doubling
    print("independent audit of %d Rocket2D SC sub-problems along the device path (K = %d; trust-region weight doubled %d times on the way): relative "
          "objective gaps %s" % (len(gaps), Kn, doubled, ["%.1e" % g for g in gaps]))


def test_emu_rocket2d_scvx_path_sub_problems_against_independent_cutting_planes(emu_lib, tmp_path):
    import shutil

    import generate_rocket2d_cut_goldens as R
    import generate_subproblem_cut_goldens as C

    Kn = 15
    cfg = tmp_path / "config"
    shutil.copytree(os.path.join(os.path.dirname(scpp_amd.__file__), "config"), cfg)
    p = cfg / "Rocket2D" / "SCvx.info"
    t = p.read_text()
    assert "nondimensionalize                   false" in t
    p.write_text(t.replace("nondimensionalize                   false", "nondimensionalize                   true"))
    sc = R.scenario(True)
    ms, rs = sc["m_scale"], sc["r_scale"]
    m = scpp_amd.Rocket2D(str(cfg)).loadParameters()
    alg = scpp_amd.SCvxAlgorithm(m, K=Kn, batch_max=1, library=emu_lib).initialize()
    alpha = float(alg.opts.alpha)
    path = _device_path(alg, m.x_init[None], 5)
    alg.ctx.close()
    Xb, Ub = _nondim2d(path[0]["X"], path[0]["U"], ms, rs)
    r_prev, solves_prev, gaps, rejected = float(path[0]["trust_region"]), 0, [], 0
    for j, st in enumerate(path[1:]):
        if st["sc_iters"] <= j:  # converged before this cap
            break
        n_rej = int(st["solves"] - solves_prev) - 1
        assert n_rej >= 0
        rejected += n_rej
        r_used = r_prev / alpha ** n_rej
        Xc, Uc = _nondim2d(st["X"], st["U"], ms, rs)
        dd = R.discretize(sc, Kn, Xb, Ub, sc["final_time"], False)
        pb = R.SubProblem(sc, Kn, Xb, Ub, sc["final_time"], dd, "scvx", dict(vc=W_VC, tr=r_used))
        _, info = C.solve_cuts(pb, verbose=False)
        assert info["cone_violation"] <= 1e-9 and max(info["eq_violation"], info["lin_violation"]) <= 1e-7
        v, l1 = _r2d_candidate(R, pb, dd, Kn, Xc, Uc, sc["final_time"], Xb, Ub, sc["final_time"])
        gap = (pb.cost(v) - info["objective"]) / max(info["objective"], 1e-9)
        assert -1e-6 <= gap <= 5e-5, ("Rocket2D SCvx", j + 1, pb.cost(v), info["objective"])
        gaps.append(gap)
        Xb, Ub, r_prev, solves_prev = Xc, Uc, float(st["trust_region"]), int(st["solves"])
    assert len(gaps) >= 3
    print("independent audit of %d Rocket2D SCvx sub-problems along the device path (K = %d, %d rejected candidates): relative objective gaps %s" % (
        len(gaps), Kn, rejected, ["%.1e" % g for g in gaps]))


# ---------------------------------------------------------------------------------------------------------------------
# SCvx with ZERO-ORDER-HOLD inputs (on the device since round 4): K - 1 inputs, no C (SCvxProblem.cpp:32-35), the trust region over the K - 1
# inputs (:58-68), the model's final-input rows on column K - 2 (rocketQuat.cpp:109-111: v_U.cols() - 1), the discretisation with a constant
# input and dV_B = Phi^-1 B (discretizationImplementation.hpp:45-52,95-100) -- restated here (forward sensitivities: P_B' = A P_B + B).
# ---------------------------------------------------------------------------------------------------------------------
class _ZohSubProblem:
    mode = "scvx"

    def __init__(self, G, sc, Xb, Ub, dd, w):
        self.G, self.sc, self.Xb, self.Ub, self.dd, self.w = G, sc, Xb, Ub, dd, w
        self.nX, self.nU, self.nN = K * 14, (K - 1) * 4, (K - 1) * 14
        self.oU = self.nX; self.oP = self.oU + self.nU; self.oM = self.oP + self.nN
        self.n = self.oM + self.nN

    def split(self, v):
        return (v[:self.nX].reshape(K, 14), v[self.oU:self.oP].reshape(K - 1, 4), v[self.oP:self.oM].reshape(K - 1, 14), v[self.oM:].reshape(K - 1, 14))

    def cost_grad(self, v):
        g = np.zeros(self.n); g[self.oP:] = self.w["vc"]
        return g

    def cost(self, v):
        return float(self.cost_grad(v) @ v)

    def eq(self, v):
        X, U, P, M = self.split(v)
        A, B, Z = self.dd
        ff = self.G.FINAL_FIXED
        r = [X[0] - self.sc["x_init"], X[K - 1][ff] - self.sc["x_final"][ff], U[K - 2][[0, 1, 3]], X[1:K - 1, 13], U[:K - 2, 3]]
        for k in range(K - 1):
            r.append(X[k + 1] - (A[k] @ X[k] + B[k] @ U[k] + Z[k] + P[k] - M[k]))
        return np.concatenate(r)

    def _forms(self):
        if hasattr(self, "_lin"):
            return self._lin, self._soc
        sc = self.sc
        iX = lambda k, j: k * 14 + j  # noqa: E731
        iU = lambda k, j: self.oU + k * 4 + j  # noqa: E731
        lin, soc = [], []
        for i in range(2 * self.nN):
            lin.append((0.0, [(self.oP + i, 1.0)]))
        for k in range(K):
            lin.append((-sc["x_final"][0], [(iX(k, 0), 1.0)]))  # mass >= m_dry
        for k in range(1, K - 1):  # (nodes 0 and K-1 are fixed by the equalities)
            soc.append(((0.0, [(iX(k, 3), sc["gs"])]), [(0.0, [(iX(k, 1), 1.0)]), (0.0, [(iX(k, 2), 1.0)])]))
            lin.append((0.0, [(iX(k, 3), 1.0)]))
            soc.append(((sc["tilt"], []), [(0.0, [(iX(k, 8), 1.0)]), (0.0, [(iX(k, 9), 1.0)])]))
            soc.append(((sc["wmax"], []), [(0.0, [(iX(k, 11 + j), 1.0)]) for j in range(3)]))
        for k in range(K - 1):
            lin.append((-sc["T_min"], [(iU(k, 2), 1.0)]))
            soc.append(((sc["T_max"], []), [(0.0, [(iU(k, j), 1.0)]) for j in range(3)]))
            soc.append(((0.0, [(iU(k, 2), sc["gim"])]), [(0.0, [(iU(k, 0), 1.0)]), (0.0, [(iU(k, 1), 1.0)])]))
            soc.append(((self.w["tr"], []), [(-self.Ub[k, j], [(iU(k, j), 1.0)]) for j in range(4)]))
        self._lin, self._soc = lin, soc
        return lin, soc

    def ineq(self, v):
        lin, soc = self._forms()
        val = lambda f: f[0] + sum(cf * v[i] for i, cf in f[1])  # noqa: E731
        return np.array([val(f) for f in lin] + [val(t) - np.sqrt(sum(val(w) ** 2 for w in ws)) for t, ws in soc])


def _discretize_zoh(sc, X, U, sigma):
    import sympy as sp
    from scipy.integrate import solve_ivp
    from generate_goldens import rocketquat_sym

    x_, u_, p_, f_ = rocketquat_sym()
    fn = sp.lambdify([x_, u_, p_], [f_, f_.jacobian(sp.Matrix(x_)), f_.jacobian(sp.Matrix(u_))], "numpy")
    dt = sigma / (K - 1)
    A = np.zeros((K - 1, 14, 14)); B = np.zeros((K - 1, 14, 4)); Z = np.zeros((K - 1, 14))
    for k in range(K - 1):
        def rhs(tau, y):
            x = y[:14]; Phi = y[14:210].reshape(14, 14); PB = y[210:266].reshape(14, 4); pz = y[266:280]
            fx, a, b = fn(x, U[k], sc["par"])
            fx = np.asarray(fx, dtype=float).ravel(); a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
            return np.concatenate([fx, (a @ Phi).ravel(), (a @ PB + b).ravel(), a @ pz - a @ x - b @ U[k] + fx])
        y = solve_ivp(rhs, [0, dt], np.concatenate([X[k], np.eye(14).ravel(), np.zeros(56 + 14)]), method="DOP853", rtol=1e-13, atol=1e-16).y[:, -1]
        A[k] = y[14:210].reshape(14, 14); B[k] = y[210:266].reshape(14, 4); Z[k] = y[266:280]
    return A, B, Z


def test_emu_scvx_zero_order_hold_path_against_independent_cutting_planes(emu_lib, tmp_path):
    import shutil

    import generate_subproblem_cut_goldens as C
    import generate_subproblem_goldens as G

    G.K = K
    sc = G.scenario()
    ms, rs = sc["m_scale"], sc["r_scale"]
    cfg = tmp_path / "config"
    shutil.copytree(os.path.join(os.path.dirname(scpp_amd.__file__), "config"), cfg)
    p = cfg / "RocketQuat" / "SCvx.info"
    t = p.read_text()
    assert "interpolate_input                   true" in t and "nondimensionalize                   true" in t
    p.write_text(t.replace("interpolate_input                   true", "interpolate_input                   false"))
    m = scpp_amd.RocketQuat(str(cfg)).loadParameters()
    alg = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=1, library=emu_lib).initialize()
    assert alg.opts.interpolate_input == 0
    alpha = float(alg.opts.alpha)
    path = _device_path(alg, m.x_init[None], 4)
    alg.ctx.close()
    Xb, Ub = _nondim(path[0]["X"], path[0]["U"], ms, rs)
    r_prev, solves_prev, gaps, rejected = float(path[0]["trust_region"]), 0, [], 0
    for j, st in enumerate(path[1:]):
        n_rej = int(st["solves"] - solves_prev) - 1
        assert st["sc_iters"] == j + 1 and n_rej >= 0
        rejected += n_rej
        r_used = r_prev / alpha ** n_rej
        Xc, Uc = _nondim(st["X"], st["U"], ms, rs)
        assert (Uc[K - 1] == 0).all()  # K - 1 inputs: the slot of node K - 1 is unused
        dd = _discretize_zoh(sc, Xb, Ub[:K - 1], sc["final_time"])
        pb = _ZohSubProblem(G, sc, Xb, Ub[:K - 1], dd, dict(vc=W_VC, tr=r_used))
        _, info = C.solve_cuts(pb, verbose=False)
        assert info["cone_violation"] <= 1e-9 and max(info["eq_violation"], info["lin_violation"]) <= 1e-7
        A, B, Z = dd
        nu = np.array([Xc[k + 1] - (A[k] @ Xc[k] + B[k] @ Uc[k] + Z[k]) for k in range(K - 1)])
        v = np.concatenate([Xc.ravel(), Uc[:K - 1].ravel(), np.maximum(nu, 0).ravel(), np.maximum(-nu, 0).ravel()])
        assert np.abs(pb.eq(v)).max() <= 1e-9 and pb.ineq(v).min() >= -1e-9, (j + 1, float(np.abs(pb.eq(v)).max()), float(pb.ineq(v).min()))
        gap = (pb.cost(v) - info["objective"]) / info["objective"]
        assert -1e-6 <= gap <= 5e-5, ("ZOH objective", j + 1, pb.cost(v), info["objective"])
        gaps.append(gap)
        Xb, Ub, r_prev, solves_prev = Xc, Uc, float(st["trust_region"]), int(st["solves"])
    print("independent audit of %d zero-order-hold SCvx sub-problems along the device path (K = %d, %d rejected candidates): relative objective gaps %s" % (
        len(gaps), K, rejected, ["%.1e" % g for g in gaps]))


def test_emu_simulate_against_dop853_of_the_restated_flow_maps(emu_lib):
    """scpp_hip_simulate (simulation.cpp:25-42: RKF78, 20 fixed steps, first-order-hold input) against DOP853 (rtol 1e-13) on the sympy restatements of
    both models' flow maps (generate_goldens.py: rocketquat_sym; generate_rocket2d_cut_goldens.py: flow_map) -- the plant step of SC_sim / MPC_sim."""
    import sympy as sp
    from scipy.integrate import solve_ivp

    import generate_rocket2d_cut_goldens as R
    import generate_subproblem_goldens as G
    from generate_goldens import rocketquat_sym

    rng = np.random.default_rng(5)
    # RocketQuat, nondimensional scenario parameters of the restatement
    G.K = K
    sc = G.scenario()
    x_, u_, p_, f_ = rocketquat_sym()
    fq = sp.lambdify([x_, u_, p_], f_, "numpy")
    m = scpp_amd.RocketQuat().loadParameters()
    ctx = scpp_amd.Context(scpp_amd.MODEL_ROCKETQUAT, K=K, batch_max=4, library=emu_lib)
    ctx.set_flow_params(np.tile(sc["par"], (4, 1)))
    X0, U0, _ = G.initial_trajectory(sc)
    x0 = X0[[1, 4, 7, 10]] + 1e-2 * rng.standard_normal((4, 14))
    u0 = U0[:4] + 1e-2 * rng.standard_normal((4, 4)); u1 = U0[:4] + 1e-2 * rng.standard_normal((4, 4))
    u0[:, 3] = 0.0; u1[:, 3] = 0.0  # no roll torque (rocketQuat.cpp:141-142; with J_z = 4e-6 in these units 1e-2 of it would spin the body up to 240 rad/s)
    dt = np.array([0.3, 0.8, 1.1, 0.05])
    xd = ctx.simulate(dt, u0, u1, x0)
    for b in range(4):
        rhs = lambda t, x: np.asarray(fq(x, u0[b] + t / dt[b] * (u1[b] - u0[b]), sc["par"]), dtype=float).ravel()  # noqa: E731
        xe = solve_ivp(rhs, [0, dt[b]], x0[b], method="DOP853", rtol=1e-13, atol=1e-16).y[:, -1]
        assert np.abs(xe - xd[b]).max() <= 1e-10 * max(1.0, np.abs(xe).max()), ("RocketQuat", b, float(np.abs(xe - xd[b]).max()))
    ctx.close()
    # Rocket2D, SI units as shipped
    s2 = R.scenario(False)
    f2 = R.flow_map()
    ctx = scpp_amd.Context(scpp_amd.MODEL_ROCKET2D, K=K, batch_max=3, library=emu_lib)
    ctx.set_flow_params(np.tile(s2["par"], (3, 1)))
    x0 = np.array([s2["x_init"]] * 3) * (1 + 1e-2 * rng.standard_normal((3, 6)))
    u0 = np.array([[0.05, 2.0e5], [-0.1, 3.0e5], [0.0, 1.0e5]]); u1 = np.array([[-0.05, 2.5e5], [0.1, 1.0e5], [0.2, 4.0e5]])
    dt = np.array([0.4, 1.0, 0.01])
    xd = ctx.simulate(dt, u0, u1, x0)
    for b in range(3):
        rhs = lambda t, x: np.asarray(f2(x, u0[b] + t / dt[b] * (u1[b] - u0[b]), s2["par"])[0], dtype=float).ravel()  # noqa: E731
        xe = solve_ivp(rhs, [0, dt[b]], x0[b], method="DOP853", rtol=1e-13, atol=1e-16).y[:, -1]
        assert np.abs(xe - xd[b]).max() <= 1e-10 * max(1.0, np.abs(xe).max()), ("Rocket2D", b, float(np.abs(xe - xd[b]).max()))
    ctx.close()


@pytest.mark.gpu
def test_sc_rocket2d_and_zero_order_hold_paths_against_independent_cutting_planes_on_gpu(hip_lib, tmp_path):
    """the path audits that ran on the CPU wave emulator only -- the SC mode (free final time, soft trust region), the second
    model in both modes, SCvx with zero-order-hold inputs -- on the device the product runs on: the same functions, the HIP library."""
    _sc_path_audit(hip_lib, 4)
    test_emu_rocket2d_sc_path_sub_problems_against_independent_cutting_planes(hip_lib)
    for name, fn in (("r2d", test_emu_rocket2d_scvx_path_sub_problems_against_independent_cutting_planes),
                     ("zoh", test_emu_scvx_zero_order_hold_path_against_independent_cutting_planes)):
        d = tmp_path / name
        d.mkdir()
        fn(hip_lib, d)
