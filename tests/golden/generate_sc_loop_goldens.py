"""Independent pin of the WHOLE SCAlgorithm loop on the shipped RocketQuat scenario.

The shipped scenario never meets SCAlgorithm's convergence test (sum(delta) < delta_tol and ||nu||_1 < nu_tol,
scpp_core/src/SCAlgorithm.cpp:131): the iteration stalls at a fixed point with ||nu||_1 ~ 0.02 .. 0.1, so every instance runs all
15 iterations and `converged_fraction` is 0 in SC mode.  That statement used to rest on the build's own two solvers.  Here the
loop of SCAlgorithm::solve / iterate (SCAlgorithm.cpp:66-189: discretise at the current iterate, solve the sub-problem, take its
solution as the next iterate, double the trust-region weight whenever ||nu||_1 < nu_tol, stop on the convergence test or after
max_iterations) is driven with scipy's trust-constr on the NLP restatement of generate_subproblem_goldens.py -- no oracle, no HIP
library, no scpp_amd -- at K = 5 and at the reference's shipped K = 15, and ||nu||_1, sum(delta), sigma, delta_sigma and the
objective of every iteration are recorded in rocketquat_sc_loop_K{5,15}.npz.

  python tests/golden/generate_sc_loop_goldens.py 5 15        (K = 15: ~680 variables per sub-problem, tens of minutes)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate_subproblem_goldens as G

NU_TOL, DELTA_TOL, MAX_ITERATIONS = 1e-5, 1e-3, 15  # SC.info:8-16
WEIGHTS = dict(t=1.0, trt=1.0, trx=50.0, vc=1000.0)


def run(K):
    G.K = K  # the restatement reads its horizon from the module global
    sc = G.scenario()
    Xb, Ub, sb = G.initial_trajectory(sc)
    w = dict(WEIGHTS)
    rec = dict(norm1_nu=[], sum_delta=[], sigma=[], delta_sigma=[], objective=[], weight_trx=[], constr_violation=[], X=[], U=[])
    converged = False
    for it in range(1, MAX_ITERATIONS + 1):
        t0 = time.time()
        dd = G.discretize(sc, Xb, Ub, sb, True)
        pb = G.SubProblem(sc, Xb, Ub, sb, dd, "sc", dict(w))
        v, r = pb.solve()
        X, U, P, M, D, sig, dsg = pb.split(v)
        n1, sd = float((P + M).sum()), float(D.sum())
        rec["norm1_nu"].append(n1); rec["sum_delta"].append(sd); rec["sigma"].append(float(sig)); rec["delta_sigma"].append(float(dsg))
        rec["objective"].append(float(r.fun)); rec["weight_trx"].append(w["trx"]); rec["constr_violation"].append(float(r.constr_violation))
        rec["X"].append(X.copy()); rec["U"].append(U.copy())
        print("K=%d iteration %2d: ||nu||_1 %.9f  sum(delta) %.3e  sigma %.9f  obj %.9f  (%.0f s)" % (K, it, n1, sd, sig, r.fun, time.time() - t0), flush=True)
        Xb, Ub, sb = X.copy(), U.copy(), float(sig)  # readSolution: the solution is the next linearisation point (SCAlgorithm.cpp:100,191-210)
        if n1 < NU_TOL:
            w["trx"] *= 2.0  # SCAlgorithm.cpp:112-115
        if sd < DELTA_TOL and n1 < NU_TOL:  # SCAlgorithm.cpp:131
            converged = True
            break
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(K=K, iterations=len(rec["sigma"]), converged=int(converged), x_init=sc["x_init"], m_scale=sc["m_scale"], r_scale=sc["r_scale"])
    np.savez(os.path.join(HERE, "rocketquat_sc_loop_K%d.npz" % K), **out)
    print("written rocketquat_sc_loop_K%d.npz: %d iterations, converged %d" % (K, out["iterations"], int(converged)))


if __name__ == "__main__":
    for k in [int(a) for a in sys.argv[1:]] or [5]:
        run(k)
