"""The two oracle solvers: known-answer SOCPs, KKT conditions, and literal-vs-structured agreement."""
import numpy as np


def test_known_answer_soc(oracle):
    # min x0 s.t. ||(x1,x2)|| <= x0, x1 + x2 = 2   ->  x = (sqrt2, 1, 1)
    r = oracle.socp_solve([1, 0, 0], [[0, 1, 1]], [2], -np.eye(3), np.zeros(3), 0, [3])
    assert r["exitflag"] == 0
    assert np.abs(r["x"] - [np.sqrt(2), 1, 1]).max() < 1e-7


def test_known_answer_lp_and_kkt(oracle):
    # min -x0 - 2 x1  s.t. x0 + x1 <= 4, x0 <= 3, x1 <= 2, x >= 0   -> x = (2,2), cost -6
    c = np.array([-1.0, -2.0])
    G = np.array([[1, 1], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=float)
    h = np.array([4, 3, 2, 0, 0], dtype=float)
    r = oracle.socp_solve(c, np.zeros((0, 2)), [], G, h, 5, [])
    assert r["exitflag"] == 0
    assert np.abs(r["x"] - [2, 2]).max() < 1e-6
    # KKT: primal feasibility, dual feasibility, complementarity
    assert np.abs(G @ r["x"] + r["s"] - h).max() < 1e-7
    assert np.abs(c + G.T @ r["z"]).max() < 1e-7
    assert (r["s"] > -1e-9).all() and (r["z"] > -1e-9).all() and abs(r["s"] @ r["z"]) < 1e-6


def test_random_socps_satisfy_kkt(oracle):
    rng = np.random.default_rng(5)
    for trial in range(5):
        n, p, l, q = 8, 2, 3, [4, 3]
        m = l + sum(q)
        # strictly feasible primal-dual pair by construction
        x0 = rng.normal(size=n)
        s0 = np.concatenate([rng.uniform(0.5, 2, l)] + [np.concatenate([[2.0 + np.linalg.norm(v)], v]) for v in (rng.normal(size=qi - 1) for qi in q)])
        z0 = np.concatenate([rng.uniform(0.5, 2, l)] + [np.concatenate([[2.0 + np.linalg.norm(v)], v]) for v in (rng.normal(size=qi - 1) for qi in q)])
        y0 = rng.normal(size=p)
        G = rng.normal(size=(m, n)); A = rng.normal(size=(p, n))
        h = G @ x0 + s0; b = A @ x0; c = -(A.T @ y0 + G.T @ z0)
        r = oracle.socp_solve(c, A, b, G, h, l, q)
        assert r["exitflag"] == 0
        assert np.abs(A @ r["x"] - b).max() < 1e-6
        assert np.abs(G @ r["x"] + r["s"] - h).max() < 1e-6
        assert np.abs(c + A.T @ r["y"] + G.T @ r["z"]).max() < 1e-6
        assert abs(r["s"] @ r["z"]) < 1e-5
        assert abs(r["pcost"] - r["dcost"]) < 1e-5 * max(1, abs(r["pcost"]))


def test_literal_problem_dimensions_match_survey(oracle):
    """n=2325, p=814, l=1474, 301 cones, m=3277 for RocketQuat K=50."""
    sc = oracle.SC(oracle.ROCKETQUAT, K=50)
    sc.solve()  # literal standard form (default solver)
    m = sc.meta()
    assert (m["n"], m["p"], m["l"], m["ncones"], m["m"]) == (2325, 814, 1474, 301, 3277)


def test_structured_and_literal_solvers_agree_on_subproblems(oracle):
    """Independent formulations (literal ECOS-style standard form vs presolved structured IPM) reach the same
    sub-problem optimum on the first SC iterations of the shipped scenario."""
    a = oracle.SC(oracle.ROCKETQUAT, K=20); a.set_solver(0); a.solve()
    b = oracle.SC(oracle.ROCKETQUAT, K=20); b.set_solver(1); b.solve()
    ia, ib = a.info(), b.info()
    for it in range(2):
        Xa, Ua, ta = a.iterate(it + 1)
        Xb, Ub, tb = b.iterate(it + 1)
        assert abs(ia[it, 0] - ib[it, 0]) < 2e-6 * max(1, abs(ib[it, 0]))   # norm1_nu
        assert abs(ta - tb) < 5e-6 * tb  # (2.4e-6 since the twin takes primal and dual step lengths of their own, round 6: both stop at reltol 1e-7 / feastol 1e-8)
        assert np.abs(Xa - Xb).max() < 5e-6
        assert np.abs(Ua - Ub).max() < 5e-6
