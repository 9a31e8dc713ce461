#!/bin/bash
# round-5 session 18: where does the persistent kernel pay?  batch entry points (SCvx and SC, cold) at small and large batches, both engines, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import time, numpy as np, scpp_amd
from scpp_amd import _lib
m = scpp_amd.RocketQuat().loadParameters()
for B in (256, 512, 1024, 2048, 4096):
    x0 = m.randomized_initial_states(B, first=777)
    for name, eng in (("launch rounds", _lib.STREAM_POOLS), ("persistent", _lib.STREAM_PERSISTENT)):
        v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=B).initialize(); v.ctx.set_stream_engine(eng)
        v.solve(x0[:min(B, 64)])
        ts = []
        for rep in range(2):
            t0 = time.perf_counter(); n = v.solve(x0); ts.append(time.perf_counter() - t0)
        v.ctx.close()
        a = scpp_amd.SCAlgorithm(m, K=50, batch_max=B).initialize(); a.ctx.set_stream_engine(eng)
        a.solve(x0[:min(B, 64)])
        tc = []
        for rep in range(2):
            t0 = time.perf_counter(); a.solve(x0); tc.append(time.perf_counter() - t0)
        a.ctx.close()
        print("B %4d %-13s: SCvx batch solve %.3f s (%5.0f converged/s)   SC cold solve %.3f s (%5.0f /s)" % (B, name, min(ts), n / min(ts), min(tc), B / min(tc)))
PY
