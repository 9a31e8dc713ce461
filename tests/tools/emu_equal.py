"""Bitwise comparison of two CPU-emulator builds of the kernel sources on the same SC / SCvx instances, both models (dev container).
usage: emu_equal.py base.so new.so   -- the regression check of refactors that must not change a single bit (recompute instead of
store / reload): prints per entry point and field whether the outputs are identical and the largest difference otherwise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd

def runs(lib):
    out = {}
    m = scpp_amd.RocketQuat().loadParameters()
    x0 = m.randomized_initial_states(6)
    a = scpp_amd.SCAlgorithm(m, K=10, batch_max=6, library=lib).initialize()
    a.solve(x0); out["rq_sc"] = a.getSolution(); a.ctx.close()
    v = scpp_amd.SCvxAlgorithm(m, K=12, batch_max=6, library=lib, max_iterations=10).initialize()
    v.solve(x0); out["rq_scvx"] = v.getSolution()
    v.solveStream(x0, slots=4, pools=2); out["rq_scvx_stream"] = v.getStreamSolution(); v.ctx.close()
    m2 = scpp_amd.Rocket2D().loadParameters()
    x2 = m2.randomized_initial_states(4)
    a = scpp_amd.SCAlgorithm(m2, K=10, batch_max=4, library=lib).initialize()
    a.solve(x2); out["r2d_sc"] = a.getSolution(); a.ctx.close()
    v = scpp_amd.SCvxAlgorithm(m2, K=8, batch_max=4, library=lib, max_iterations=5).initialize()
    v.solve(x2); out["r2d_scvx"] = v.getSolution(); v.ctx.close()
    return out

A, B = runs(sys.argv[1]), runs(sys.argv[2])
bad = 0
for name in A:
    rep = {}
    for k in A[name]:
        a, b = np.asarray(A[name][k]), np.asarray(B[name][k])
        if a.dtype.kind not in "fiu":
            continue
        same = bool(np.array_equal(a, b))
        rep[k] = True if same else float(np.nanmax(np.abs(a.astype(float) - b.astype(float))))
        bad += not same
    print(name, "IDENTICAL" if all(v is True for v in rep.values()) else {k: v for k, v in rep.items() if v is not True},
          "ipm iterations", int(np.sum(A[name]["ipm_iters"])), "/", int(np.sum(B[name]["ipm_iters"])))
sys.exit(1 if bad else 0)
