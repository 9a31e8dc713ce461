"""Bitwise comparison of the interior-point schedules of ONE library on the GPU (resident kernel vs the two-kernel split, csrc/ipm_split.h)
on the same SCvx and SC instances, plus the time of each.  usage: split_equal.py [lib.so] [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
from scpp_amd import _lib
lib = os.path.join(ROOT, sys.argv[1]) if len(sys.argv) > 1 else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = scpp_amd.RocketQuat().loadParameters()
x0 = m.randomized_initial_states(B)
res = {}
for name, sched in (("resident", _lib.IPM_RESIDENT), ("split", _lib.IPM_SPLIT)):
    v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=B, library=lib).initialize()
    v.ctx.set_ipm_schedule(sched)
    v.solve(x0[:64])  # warm-up (module load)
    t0 = time.time(); v.solve(x0); t1 = time.time() - t0
    ov = v.getSolution(); v.ctx.close()
    a = scpp_amd.SCAlgorithm(m, K=50, batch_max=B, library=lib).initialize()
    a.ctx.set_ipm_schedule(sched)
    t0 = time.time(); a.solve(x0); t2 = time.time() - t0
    oa = a.getSolution(); a.ctx.close()
    res[name] = (ov, oa)
    print(f"{name:9s} SCvx batch {B}: {t1:.3f} s ({int(ov['converged'].sum())} converged, {int(ov['ipm_iters'].sum())} ipm iterations, status!=0: {int((ov['status'] != 0).sum())})   SC: {t2:.3f} s ({int(oa['ipm_iters'].sum())} ipm iterations)")
ok = True
for mode, i, keys in (("SCvx", 0, ("X", "U", "sigma", "nu_norm", "ipm_iters", "status", "sc_iters", "solves", "converged")), ("SC", 1, ("X", "U", "sigma", "nu_norm", "ipm_iters", "status", "sc_iters"))):
    a, b = res["resident"][i], res["split"][i]
    eq = {k: bool(np.array_equal(np.ascontiguousarray(a[k]).view(np.uint8), np.ascontiguousarray(b[k]).view(np.uint8))) for k in keys}
    ok &= all(eq.values())
    print(mode, "bitwise equal:", eq)
print("SPLIT_EQUAL_OK" if ok else "SPLIT_EQUAL_FAILED")
sys.exit(0 if ok else 1)
