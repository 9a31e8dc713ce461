"""Streaming engines of ONE library on the GPU: the persistent kernel (one launch, a wavefront per slot) against the pool engine (rounds of
launches) -- result rows bitwise, and the time of each.  usage: engine_equal.py [N] [slots]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
from scpp_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = scpp_amd.RocketQuat().loadParameters()
x0 = m.randomized_initial_states(N, first=5000)
rows = {}
for name, eng in (("pools", _lib.STREAM_POOLS), ("persistent", _lib.STREAM_PERSISTENT)):
    v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=S).initialize()
    v.ctx.set_stream_engine(eng)
    v.solveStream(x0[:64], slots=min(S, 64))  # warm-up
    t0 = time.time(); n = v.solveStream(x0, slots=S); dt = time.time() - t0
    rows[name] = v.ctx.stream_download_rows()
    g = scpp_amd.Context.unpack_stream_rows(rows[name], 50)
    print(f"{name:10s} N={N} slots={S}: {dt:.3f} s, {n} converged ({n / dt:.0f}/s), status!=0: {int((g['status'] != 0).sum())}, ipm iterations {int(g['ipm_iters'].sum())}, profile {v.ctx.stream_profile()}")
    v.ctx.close()
a, b = rows["pools"], rows["persistent"]
eq = a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
if not eq:
    bad = np.where((a.view(np.uint64) != b.view(np.uint64)).any(axis=1))[0]
    print("rows that differ:", len(bad), bad[:20])
    ga, gb = scpp_amd.Context.unpack_stream_rows(a, 50), scpp_amd.Context.unpack_stream_rows(b, 50)
    for k in ("sc_iters", "solves", "converged", "status", "ipm_iters"):
        print(k, int((ga[k] != gb[k]).sum()))
    print("max |dX|", float(np.abs(ga["X"] - gb["X"]).max()))
print("ENGINE_EQUAL_OK" if eq else "ENGINE_EQUAL_FAILED")
sys.exit(0 if eq else 1)
