#!/bin/bash
# round-5 session 17: SC_sim-shaped work (4096 instances, warm-started solves back to back) -- persistent SC kernel vs the loop of launches, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import time, numpy as np, scpp_amd
from scpp_amd import _lib
m = scpp_amd.RocketQuat().loadParameters()
for B in (4096, 1024, 8192):
    x0 = m.randomized_initial_states(B, first=123)
    for rep in range(2):
        for name, eng in (("loop of launches", _lib.STREAM_POOLS), ("persistent", _lib.STREAM_PERSISTENT)):
            a = scpp_amd.SCAlgorithm(m, K=50, batch_max=B).initialize(); a.ctx.set_stream_engine(eng)
            a.solve(x0); a.ctx.synchronize()
            t0 = time.perf_counter()
            for s in range(10):
                a.solve(x0 * (1.0 + 1e-4 * s), warm_start=True)
            a.ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
            o = a.getSolution()
            print("B %d warm-started SC solve, %-16s: %.1f ms per solve of the batch (%.0f solves/s), ipm iterations per instance %.1f" % (B, name, 1e3 * dt, B / dt, o["ipm_iters"].mean()))
            a.ctx.close()
PY
