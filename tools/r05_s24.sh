#!/bin/bash
# round-5 session 24: the blow-up guard -- instance 8392 and the default bench job's instances
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import numpy as np, scpp_amd
from scpp_amd import _lib
m = scpp_amd.RocketQuat().loadParameters()
x1 = m.randomized_initial_states(1, first=8392)
for engine, name in ((_lib.STREAM_POOLS, "rounds"), (_lib.STREAM_PERSISTENT, "persistent")):
    a = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=1).initialize(); a.ctx.set_stream_engine(engine)
    a.solve(x1); s = a.getSolution()
    print("instance 8392 (%s): status %d converged %d sc_iters %d solves %d ipm %d nu %.4e" % (name, s["status"][0], s["converged"][0], s["sc_iters"][0], s["solves"][0], s["ipm_iters"][0], s["nu_norm"][0]))
    a.ctx.close()
B = 8192
v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=B).initialize()
x0 = np.concatenate([m.randomized_initial_states(B, first=8192 * i) for i in range(1, 5)])
n = v.solveStream(x0, slots=B); o = v.getStreamSolution()
print("default bench job's instances (8192 .. 40959): converged %d of %d, status != 0: %d, mean ipm %.2f" % (n, len(x0), int((o["status"] != 0).sum()), o["ipm_iters"].mean()))
PY
timeout 300 python tests/tools/engine_equal.py 3000 1024 2>&1 | tail -3 | cut -c1-200
