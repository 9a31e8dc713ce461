#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import numpy as np, scpp_amd
from scpp_amd import _lib
m = scpp_amd.RocketQuat().loadParameters()
x1 = m.randomized_initial_states(1, first=8392)
for maxit in (1, 2):
    a = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=1, max_iterations=maxit).initialize(); a.ctx.set_stream_engine(_lib.STREAM_POOLS)
    a.solve(x1); s = a.getSolution(); info = a.ctx.socp_info()[0]
    print("max_it", maxit, "status", s["status"][0], "ipm", s["ipm_iters"][0], "dbg pcost %.6e gap %.6e pres %.3e dres %.3e iter %g status %g n1 %.6e sumdelta %.3e" % tuple(info[:8]), "max|U| %.3e" % np.abs(s["U"]).max())
    a.ctx.close()
# the same solve with fewer interior-point iterations allowed: where does it break?
for cap in (14, 16, 17, 18):
    a = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=1, max_iterations=2).initialize(); a.ctx.set_stream_engine(_lib.STREAM_POOLS)
    a.ctx.set_socp_opts(maxit=cap)
    a.solve(x1); s = a.getSolution(); info = a.ctx.socp_info()[0]
    print("ipm cap", cap, "status", s["status"][0], "ipm", s["ipm_iters"][0], "dbg pcost %.6e gap %.6e pres %.3e dres %.3e iter %g status %g n1 %.6e" % tuple(info[:7]), "max|U| %.3e" % np.abs(s["U"]).max())
    a.ctx.close()
PY
