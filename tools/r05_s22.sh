#!/bin/bash
# round-5 session 22: which instance of the default bench job fails its interior-point solve, and how
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import numpy as np, scpp_amd
from scpp_amd import _lib
m = scpp_amd.RocketQuat().loadParameters()
B = 8192
for first in (8192, 16384):
    x0 = m.randomized_initial_states(B, first=first)
    v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=B).initialize()
    v.solveStream(x0, slots=B)
    o = v.getStreamSolution()
    bad = np.flatnonzero(o["status"] != 0)
    print("instances", first, "..", first + B - 1, "failures:", [(int(first + i), int(o["status"][i]), int(o["sc_iters"][i]), int(o["solves"][i]), int(o["ipm_iters"][i]), float(o["trust_region"][i]), float(o["nu_norm"][i])) for i in bad])
    for i in bad:
        # replay this instance alone, iteration by iteration, with the debug record of the last solve
        a = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=1).initialize()
        for engine, name in ((_lib.STREAM_PERSISTENT, "persistent"), (_lib.STREAM_POOLS, "rounds")):
            a.ctx.set_stream_engine(engine)
            a.solve(x0[i:i + 1]); s = a.getSolution(); info = a.ctx.socp_info()[0]
            print("  alone (%s): status %d sc_iters %d solves %d ipm %d radius %.3e | last solve: pcost %.6e gap %.2e pres %.2e dres %.2e iters %d status %d" % (
                name, s["status"][0], s["sc_iters"][0], s["solves"][0], s["ipm_iters"][0], s["trust_region"][0], info[0], info[1], info[2], info[3], info[4], info[5]))
        a.ctx.close()
    v.ctx.close()
PY
