"""Bitwise comparison of two builds of the HIP library on the same SC and SCvx instances (GPU box). usage: lib_equal.py a.so b.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
m = scpp_amd.RocketQuat().loadParameters()
res = []
import scpp_amd._lib as _L
ALL_SYMBOLS, ABI_NOW = list(_L.SYMBOLS), _L.ABI_REVISION
for lib in sys.argv[1:3]:
    lib = os.path.join(ROOT, lib)
    # a library of ABI revision 6 (before the plugin registry / the third model: revision 7 only ADDED entry points and a parameter struct) is loaded
    # with the binding's checks relaxed to what it can export -- this tool compares results, not interfaces
    import ctypes
    import scpp_amd._lib as L
    probe = ctypes.CDLL(lib)
    if not hasattr(probe, "scpp_hip_sc_setup_lander3dof"):
        L.SYMBOLS = [s for s in L.SYMBOLS if "lander3dof" not in s]
        L.ABI_REVISION = 6
    else:
        L.SYMBOLS = list(ALL_SYMBOLS)
        L.ABI_REVISION = ABI_NOW
    a = scpp_amd.SCAlgorithm(m, K=50, batch_max=1024, library=lib).initialize()
    x0 = m.randomized_initial_states(1024)
    a.solve(x0); o = a.getSolution(); a.ctx.close()
    v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=64, library=lib).initialize()
    v.solve(x0[:64]); ov = v.getSolution(); v.ctx.close()
    # the streaming engine's persistent kernel (more instances than slots), and the reference's other model through the same entry points
    sx = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=256, library=lib).initialize()
    sx.solveStream(x0[:512], slots=256); osx = sx.getStreamSolution(); sx.ctx.close()
    m2 = scpp_amd.Rocket2D().loadParameters()
    x2 = m2.randomized_initial_states(128)
    a2 = scpp_amd.SCAlgorithm(m2, K=30, batch_max=128, library=lib).initialize()
    a2.solve(x2); o2 = a2.getSolution(); a2.ctx.close()
    v2 = scpp_amd.SCvxAlgorithm(m2, K=30, batch_max=128, library=lib).initialize()
    v2.solve(x2); ov2 = v2.getSolution(); v2.ctx.close()
    res.append((o, ov, osx, o2, ov2))
for name, i in (("SC", 0), ("SCvx", 1), ("SCvx stream (persistent kernel)", 2), ("Rocket2D SC", 3), ("Rocket2D SCvx", 4)):
    a, b = res[0][i], res[1][i]
    print(name, {k: bool(np.array_equal(a[k], b[k])) for k in ("X", "U", "sigma", "ipm_iters", "status", "sc_iters")})
