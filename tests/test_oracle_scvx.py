"""Oracle restatement of the SCvx variant (oracle/scvx.hpp: SCvxProblem.cpp:6-71, SCvxAlgorithm.cpp:22-278):
problem dimensions against SURVEY.md §8(a) row a8', the structured twin's SCvx mode against the literal
standard-form solver, and the accept / reject / radius-update logic."""
import numpy as np


def test_scvx_literal_problem_dimensions(oracle):
    """n = 2273, p = 814, l = 1473, 300 cones (sum dim 1100) for RocketQuat K = 50 (SURVEY a8')."""
    s = oracle.SCvx(K=50)
    s.set_solver(0)
    s.set_tolerances(1e-8, 1e-6, 1e-6, 3)  # only the assembly matters here: stop after 3 IPM iterations
    s.set_max_iterations(1)
    s.solve()
    m = s.meta()
    assert (m["n"], m["p"], m["l"], m["ncones"]) == (2273, 814, 1473, 300)
    assert m["m"] == 1473 + 1100


def test_scvx_twin_matches_literal_solver_on_first_subproblem(oracle):
    K = 12
    a = oracle.SCvx(K=K); a.set_solver(0); a.set_tolerances(1e-8, 1e-6, 1e-6, 100); a.set_max_iterations(1)
    b = oracle.SCvx(K=K); b.set_solver(1); b.set_max_iterations(1)
    assert a.solve() == 0 and b.solve() == 0
    Xa, Ua, ta = a.iterate(1)
    Xb, Ub, tb = b.iterate(1)
    assert ta == tb  # fixed final time
    # the objective (virtual control only) pins norm1_nu tightly; X and U sit on a flat optimal face, so two
    # solvers stopped at different tolerances agree there to ~1e-5
    assert np.abs(Xa - Xb).max() <= 2e-4 * np.abs(Xa).max()
    assert np.abs(Ua - Ub).max() <= 2e-4 * np.abs(Ua).max()
    ia, ib = a.info()[0], b.info()[0]
    assert abs(ia[0] - ib[0]) <= 1e-6 * abs(ia[0])          # norm1_nu
    assert abs(ia[1] - ib[1]) <= 1e-3 * abs(ia[1]) + 1e-6   # nonlinear cost of the candidate
    # the hard input trust region is respected by both
    X0, U0, _ = a.iterate(0)
    assert np.linalg.norm(Ua - U0, axis=1).max() <= 5.0 * (1 + 1e-6)
    assert np.linalg.norm(Ub - U0, axis=1).max() <= 5.0 * (1 + 1e-6)


def test_scvx_accept_reject_logic(oracle):
    """rho = dJ/dL drives the radius (SCvxAlgorithm.cpp:121-152); rejected candidates restore the trajectory and
    re-solve without re-discretising; convergence = |dL| < change_threshold."""
    s = oracle.SCvx(K=10); s.set_solver(1); s.set_max_iterations(12)
    assert s.solve() == 0
    m, info = s.meta(), s.info()
    assert m["solves"] == len(info) and m["solves"] >= m["iterations"]
    assert info[0][6] == 2  # first pass only records the nonlinear cost
    tr = 5.0
    for r in info[1:]:
        norm1, J, dJ, dL, rho, tr_after, acc = r[:7]
        if acc == 3:
            assert abs(dL) < 1e-3
            assert tr_after == tr
        elif acc == 0:
            assert rho < 0.0 and abs(tr_after - tr / 2.0) < 1e-15
        else:
            assert rho >= 0.0
            want = tr / 2.0 if rho < 0.25 else (tr * 3.2 if rho >= 0.9 else tr)
            assert abs(tr_after - want) < 1e-15
        tr = tr_after
    assert m["converged"] == int(info[-1][6] == 3)
    # the nonlinear defect of the final trajectory is what the last accepted row reports
    assert info[-1][1] < info[0][1]


def test_scvx_literal_whole_run_at_K50(oracle):
    """The literal (reference-shaped, n=2273) solver carries whole SCvx runs at K=50 since it follows ECOS's safeguards (best
    iterate kept, returned as "close to optimal" when the path breaks down near the degenerate optimum where the virtual
    control is driven onto a face of its l1 ball).  The SCvx sub-problem's cost is ||nu||_1 alone: its optimum is not unique
    in the inputs, two interior-point formulations return different points of the optimal set, the accept / reject sequences
    drift apart, and whole runs agree on the verdict, on the cost levels and to ~1e-3 on the states only."""
    a = oracle.SCvx(K=50); a.randomize(20260927, 0); a.set_solver(0)
    b = oracle.SCvx(K=50); b.randomize(20260927, 0); b.set_solver(1)
    assert a.solve() == 0 and b.solve() == 0
    ma, mb = a.meta(), b.meta()
    assert ma["converged"] == 1 and mb["converged"] == 1
    assert abs(ma["iterations"] - mb["iterations"]) <= 4
    Xa, Ua, _ = a.iterate(-1)
    Xb, Ub, _ = b.iterate(-1)
    assert np.abs(Xa - Xb).max() <= 2e-3 * np.abs(Xb).max()
    # both end at the same level of virtual control (linear cost) and nonlinear defect
    ia, ib = a.info()[-1], b.info()[-1]
    print("literal", ia[:2], "twin", ib[:2])
    assert abs(ia[0] - ib[0]) <= 0.25 * max(ia[0], ib[0])
    assert abs(ia[1] - ib[1]) <= 0.25 * max(ia[1], ib[1])


def test_native_batch_runner_matches_the_per_instance_calls(oracle):
    """oracle_scvx_run_batch (bench.py's cpu_baseline leg: native threads, one instance counter) does per instance what the Python
    wrapper does: same convergence, iteration and solve totals."""
    import ctypes as C

    n, first, K = 4, 7, 20
    counts = (C.c_longlong * 4)()
    rc = oracle.lib().oracle_scvx_run_batch(oracle.CONFIG_ROOT.encode(), K, C.c_ulonglong(20260927), C.c_ulonglong(first), n, 1, 3, counts)
    assert rc == 0
    conv = iters = solves = 0
    for i in range(n):
        s = oracle.SCvx(K=K); s.randomize(20260927, first + i); s.set_solver(1)
        assert s.solve() == 0
        m = s.meta()
        conv += m["converged"]; iters += m["iterations"]; solves += m["solves"]
    assert [int(v) for v in counts] == [conv, 0, iters, solves]
