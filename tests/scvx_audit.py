"""Literal-solver audit of the device's SCvx path (TEST INFRASTRUCTURE: uses the oracle).

Why whole-run comparisons with an independent solver cannot be the parity test of the SCvx mode.  The accept / reject rule of
SCvxAlgorithm (rho = dJ / dL against rho_0 = 0, SCvxAlgorithm.cpp:118-152) turns last-digit differences of a sub-problem optimum
into different decision sequences: the oracle's two solvers (structured twin, literal sparse-KKT) part ways at the 2nd or 3rd
solve on most instances and then visit different iterates (inputs differ by 1e-2 .. 1e-1 at the end, both runs converge).  The
reference itself, with ECOS, would take a third path.  What CAN be pinned independently is every sub-problem the device
actually solved:

  for every ACCEPTED iterate j -> j+1 of the device run: build the reference-shaped (literal) sub-problem of
  SCvxProblem.cpp:6-71 + rocketQuat.cpp:70-144 linearised at the device's iterate j with the radius the device used, and check
  that the device's iterate j+1 is (i) feasible in it, row by row, to 1e-9, and (ii) as good as the literal solver's optimum of
  the same problem (objective gap within the two solvers' termination tolerances).

The accepted iterates are obtained without any extra ABI: the engine is deterministic, so a run capped at max_iterations = j
ends exactly at iterate j of the full run (asserted: the capped runs' counters are prefixes of the full run's).
A rejected candidate is never an iterate; the radius of the accepted solve of iteration j+1 is radius_j / alpha^(#rejections).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def device_path_capped(alg, x0, max_iterations):
    """path[j] = dict(X, U, radius, solves, iters, converged) after j iterations, j = 0 .. (until every instance has ended);
    path[0] is the initial trajectory (getInitializedTrajectory, redimensionalised) with the configured radius.
    The way rounds 3 - 5 recovered the iterates: j + 1 runs capped at max_iterations = 0 .. j (the engine is deterministic, so a run capped at j ends at
    iterate j of the full run).  Kept as the independent check of device_path below (tests: bitwise the same path)."""
    path = []
    keep = alg._max_iterations
    try:
        for j in range(0, max_iterations + 1):
            alg._max_iterations = j
            alg.solve(x0)
            o = alg.getSolution()
            assert (o["status"] == 0).all()
            path.append(dict(X=o["X"].copy(), U=o["U"].copy(), radius=o["trust_region"].copy(), solves=o["solves"].copy(),
                             iters=o["sc_iters"].copy(), converged=o["converged"].copy()))
            if (o["converged"] == 1).all():
                break
    finally:
        alg._max_iterations = keep
    return path


def device_path(alg, x0, max_iterations):
    """The same path from ONE run: the device records every iterate (scpp_hip_scvx_record_iterates / _download_iterates, round 6 -- the counterpart of
    SCvxAlgorithm::getAllSolutions, SCvxAlgorithm.cpp:245-260: the initial trajectory and the trajectory after every iteration, with the radius, the
    solve count and J at that point)."""
    keep = alg._max_iterations
    try:
        alg._max_iterations = max_iterations
        alg.ctx.scvx_record_iterates(True)
        alg.solve(x0)
        o = alg.getSolution()
        assert (o["status"] == 0).all()
        X, U, n, sc = alg.ctx.scvx_iterates(max_iterations + 1)
    finally:
        alg._max_iterations = keep
        alg.ctx.scvx_record_iterates(False)
    B = X.shape[0]
    rows = np.arange(B)
    assert (n >= 1).all() and (n - 1 == np.where(max_iterations > 0, o["sc_iters"], 0)).all() and np.array_equal(X[rows, n - 1], o["X"]) and np.array_equal(U[rows, n - 1], o["U"])
    path = []
    for j in range(0, max_iterations + 1):
        jj = np.minimum(j, n - 1)  # an instance that ended earlier keeps its last iterate (what a run capped at j returns for it)
        conv = ((o["converged"] == 1) & (j >= n - 1)).astype(np.int32)
        path.append(dict(X=X[rows, jj].copy(), U=U[rows, jj].copy(), radius=sc[rows, jj, 0].copy(), solves=sc[rows, jj, 1].astype(np.int32),
                         iters=(jj if j > 0 else o["sc_iters"] * 0 + 1).astype(np.int32), converged=conv))
        if (conv == 1).all():
            break
    return path


def _dump_literal_failure(s, Xb, Ub, radius, Xc, Uc, iteration, c):
    """keep the inputs of a sub-problem on which the literal solver returned garbage, for an offline look at oracle/socp.hpp"""
    import os

    try:
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "literal_failures")
        os.makedirs(root, exist_ok=True)
        np.savez(os.path.join(root, "subproblem_it%d_%d.npz" % (iteration, len(os.listdir(root)))), Xbar=Xb, Ubar=Ub, radius=radius, Xc=Xc, Uc=Uc,
                 x_init=s.x_init() if hasattr(s, "x_init") else np.zeros(0), lit_cost=c["lit_cost"], cost=c["cost"],
                 flag=c.get("lit_exitflag_reported", 0))
    except Exception:
        pass


def audit_instance(oracle, K, seed, instance_id, path, b, alpha, lit_tol=1e-9):
    """check_point of every accepted sub-problem of instance b (= randomised RocketQuat instance `instance_id` of `seed`)."""
    s = oracle.SCvx(K=K)
    s.randomize(seed, instance_id)
    s.set_tolerances(lit_tol, lit_tol, lit_tol, 200)
    return audit_rows(s, path, b, alpha)


def audit_rows(s, path, b, alpha):
    """check_point of every accepted sub-problem of instance b of `path` with the oracle handle `s` (any model; its x_init must
    be instance b's)."""
    rows = []
    Xb, Ub = path[0]["X"][b], path[0]["U"][b]
    r_prev, solves_prev = float(path[0]["radius"][b]), 0
    for j, st in enumerate(path[1:]):
        if st["iters"][b] <= j:  # the run had ended before this cap
            break
        n_rej = int(st["solves"][b] - solves_prev) - 1
        assert n_rej >= 0
        r_used = r_prev / alpha ** n_rej
        c = s.check_point(Xb, Ub, r_used, st["X"][b], st["U"][b], True)
        c["iteration"], c["radius"], c["rejections"] = j + 1, r_used, n_rej
        ok = c["lit_exitflag"] in (0, 10)
        if ok:
            # The CHECKER is checked before it is believed: the literal solver's own point must be a point of its own problem
            # (round 4: on 1 of 540 sub-problems of a GPU session it returned exit flag 10, "close to optimal", with entries of 1e25).
            # A literal result that is not feasible to 1e-6 counts as a literal-solver FAILURE (flag -3), like its -1 / -2 exits; the
            # callers' bound on the number of such failures is unchanged.
            v = s.check_point(Xb, Ub, r_used, c["X_lit"], c["U_lit"], False)
            sane = (np.isfinite(c["lit_cost"]) and np.isfinite(v["cost"]) and v["eq_violation"] <= 1e-6 and v["min_lp_slack"] >= -1e-6
                    and v["min_cone_slack"] >= -1e-6)
            if not sane:
                c["lit_exitflag_reported"] = c["lit_exitflag"]
                c["lit_exitflag"] = -3
                ok = False
                _dump_literal_failure(s, Xb, Ub, r_used, st["X"][b], st["U"][b], j + 1, c)
        c["relX"] = float(np.abs(c["X_lit"] - st["X"][b]).max() / np.abs(st["X"][b]).max()) if ok else None
        c["relU"] = float(np.abs(c["U_lit"] - st["U"][b]).max() / np.abs(st["U"][b]).max()) if ok else None
        del c["X_lit"], c["U_lit"]
        rows.append(c)
        Xb, Ub, r_prev, solves_prev = st["X"][b], st["U"][b], float(st["radius"][b]), int(st["solves"][b])
    return rows


def audit(oracle, K, seed, first, path, alpha, instances, threads=8, lit_tol=1e-9):
    def one(b):
        return audit_instance(oracle, K, seed, first + b, path, b, alpha, lit_tol=lit_tol)

    with ThreadPoolExecutor(threads) as ex:
        per = list(ex.map(one, instances))
    rows = [r for p in per for r in p]
    solved = [r for r in rows if r["lit_exitflag"] in (0, 10)]
    gaps = np.array([(r["cost"] - r["lit_cost"]) / max(abs(r["lit_cost"]), 1e-12) for r in solved])
    return dict(
        rows=rows, n=len(rows), n_literal_solved=len(solved),
        literal_flags={f: sum(r["lit_exitflag"] == f for r in rows) for f in sorted({r["lit_exitflag"] for r in rows})},
        worst_eq=max(r["eq_violation"] for r in rows), worst_lp=min(r["min_lp_slack"] for r in rows),
        worst_cone=min(r["min_cone_slack"] for r in rows),
        gap_median=float(np.median(np.abs(gaps))), gap_max=float(np.abs(gaps).max()), gap_min_signed=float(gaps.min()),
        relX_max=max(r["relX"] for r in solved), relU_max=max(r["relU"] for r in solved),
        relU_median=float(np.median([r["relU"] for r in solved])),
    )


# ---- SC mode (SCAlgorithm): certificate for the LAST sub-problem of a device run ----
def sc_device_iterate(alg, x0, j):
    """(X, U, sigma, ...) of every instance after j SCAlgorithm iterations, redimensionalised (a run capped at
    max_iterations = j: the device loop is deterministic, so this is iterate j of the full run)."""
    import copy

    opts = copy.copy(alg.opts)
    opts.max_iterations = int(j)
    alg.ctx.sc_setup(alg.model.sc_params(), opts, np.atleast_2d(np.asarray(x0, dtype=np.float64)))
    alg.ctx.sc_solve()
    return alg.ctx.download()


def sc_last_solve_certificate(handle, alg, x0_row, iters, w_trx=0.0, lit_tol=1e-9):
    """The device's final iterate of one instance in the oracle's LITERAL SC sub-problem linearised at the device's previous
    iterate: row-by-row feasibility, objective, and the literal solver's optimum of that same problem (oracle/sc.hpp: checkPoint).
    `handle` is an oracle.SC whose x_init is x0_row."""
    prev = sc_device_iterate(alg, x0_row, iters - 1)
    last = sc_device_iterate(alg, x0_row, iters)
    handle.set_tolerances(lit_tol, lit_tol, lit_tol, 200)
    nU = handle.meta()["nU"]  # K inputs (first-order hold) or K-1 (zero-order hold: the device's slot K-1 is unused)
    Up, Ul = np.ascontiguousarray(prev["U"][0][:nU]), np.ascontiguousarray(last["U"][0][:nU])
    c = handle.check_point(prev["X"][0], Up, float(prev["sigma"][0]), last["X"][0], Ul, float(last["sigma"][0]), w_trx=w_trx)
    if c["lit_exitflag"] in (0, 10):
        c["relX"] = float(np.abs(c["X_lit"] - last["X"][0]).max() / np.abs(last["X"][0]).max())
        c["relU"] = float(np.abs(c["U_lit"] - Ul).max() / np.abs(Ul).max())
        c["gap"] = float((c["cost"] - c["lit_cost"]) / abs(c["lit_cost"]))
    c["X"], c["U"], c["sigma"] = last["X"][0], last["U"][0], float(last["sigma"][0])
    return c


def assert_certificate(c, feas=1e-9, gap=1e-6):
    """feasible in the literal problem and as good as its optimum: an eps-optimal point of the reference-shaped problem"""
    assert c["eq_violation"] <= feas and c["min_lp_slack"] >= -feas and c["min_cone_slack"] >= -feas, c
    assert c["lit_exitflag"] in (0, 10) and abs(c["gap"]) <= gap, c
