#!/bin/bash
# round-5 session 14: the persistent SC-mode kernel on hardware
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_s14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "persistent_sc_loop or full_sc_oneshot or converging_configuration" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^$" $O/pytest.log | tail -8 | cut -c1-300
python - <<'PY'
import time, numpy as np, scpp_amd
from scpp_amd import _lib
m = scpp_amd.RocketQuat().loadParameters()
B = 8192
xs = m.randomized_initial_states(2 * B, first=30_000_000)
for name, eng in (("loop of launches", _lib.STREAM_POOLS), ("persistent", _lib.STREAM_PERSISTENT), ("loop of launches", _lib.STREAM_POOLS), ("persistent", _lib.STREAM_PERSISTENT)):
    a = scpp_amd.SCAlgorithm(m, K=50, batch_max=B).initialize(); a.ctx.set_stream_engine(eng)
    a.solve(xs[:B]); a.ctx.synchronize()
    t0 = time.perf_counter(); a.solve(xs[B:]); o = a.getSolution(); dt = time.perf_counter() - t0
    print("SC mode 8192 x K=50, %s: %.0f terminated trajectories/s (ipm iterations %d)" % (name, B / dt, int(o["ipm_iters"].sum())))
    a.ctx.close()
PY
