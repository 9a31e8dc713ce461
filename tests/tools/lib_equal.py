"""Bitwise comparison of two builds of the HIP library on the same SC and SCvx instances (GPU box). usage: lib_equal.py a.so b.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
m = scpp_amd.RocketQuat().loadParameters()
res = []
for lib in sys.argv[1:3]:
    lib = os.path.join(ROOT, lib)
    a = scpp_amd.SCAlgorithm(m, K=50, batch_max=1024, library=lib).initialize()
    x0 = m.randomized_initial_states(1024)
    a.solve(x0); o = a.getSolution(); a.ctx.close()
    v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=64, library=lib).initialize()
    v.solve(x0[:64]); ov = v.getSolution(); v.ctx.close()
    res.append((o, ov))
for name, i in (("SC", 0), ("SCvx", 1)):
    a, b = res[0][i], res[1][i]
    print(name, {k: bool(np.array_equal(a[k], b[k])) for k in ("X", "U", "sigma", "ipm_iters", "status", "sc_iters")})
