#!/bin/bash
# round-5 session 23: instance 8392 -- where does the NaN of its third sub-problem come from?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import numpy as np, scpp_amd
from scpp_amd import _lib
m = scpp_amd.RocketQuat().loadParameters()
x0 = m.randomized_initial_states(1, first=8392)
for engine, name in ((_lib.STREAM_POOLS, "rounds"), (_lib.STREAM_PERSISTENT, "persistent")):
    for maxit in (1, 2, 3):
        a = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=1, max_iterations=maxit).initialize()
        a.ctx.set_stream_engine(engine)
        a.solve(x0); s = a.getSolution(); info = a.ctx.socp_info()[0]
        A, Bm, Cm, S, Z = a.ctx.download_dd()
        X, U = s["X"][0], s["U"][0]
        print("%-10s max_it %d: status %d sc_iters %d solves %d ipm %d J %.6e nu %.4e radius %.3e | X finite %s U finite %s min|T| %.3e min m %.4e | dd finite A %s B %s C %s Z %s | last solve iters %d pres %.2e" % (
            name, maxit, s["status"][0], s["sc_iters"][0], s["solves"][0], s["ipm_iters"][0], s["nonlinear_cost"][0], s["nu_norm"][0], s["trust_region"][0],
            np.isfinite(X).all(), np.isfinite(U).all(), np.linalg.norm(U[:, :3], axis=1).min(), X[:, 0].min(),
            np.isfinite(A).all(), np.isfinite(Bm).all(), np.isfinite(Cm).all(), np.isfinite(Z).all(), info[4], info[2]))
        if not np.isfinite(A).all():
            bad = np.argwhere(~np.isfinite(A[0]))
            print("   non-finite A entries at segments", sorted(set(bad[:, 0].tolist()))[:10], "rows", sorted(set(bad[:, 1].tolist())), "cols", sorted(set(bad[:, 2].tolist())))
            k = int(bad[0, 0]); print("   X[k]", X[k], "\n   U[k]", U[k], "U[k+1]", U[k + 1])
        a.ctx.close()
PY
