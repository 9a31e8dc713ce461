"""Same-box alternating A/B of the SC mode (SCAlgorithm, K = 50, bench.py's sc_mode workload) on two builds of the HIP library, plus the
bitwise comparison of their results (RocketQuat and Rocket2D).   usage: sc_mode_ab.py a.so b.so [batch] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd

libs = [os.path.join(ROOT, p) for p in sys.argv[1:3]]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
KEYS = ("X", "U", "sigma", "nu_norm", "ipm_iters", "status", "sc_iters", "converged")
for name, model, K in (("RocketQuat", scpp_amd.RocketQuat().loadParameters(), 50), ("Rocket2D", scpp_amd.Rocket2D().loadParameters(), 30)):
    Bm = B if name == "RocketQuat" else min(B, 1024)
    xs = model.randomized_initial_states(2 * Bm, seed=20260927, first=30_000_000)
    algs = [scpp_amd.SCAlgorithm(model, K=K, batch_max=Bm, library=l).initialize() for l in libs]
    outs = []
    for a in algs:
        a.solve(xs[:Bm]); a.ctx.synchronize()
    for rep in range(reps if name == "RocketQuat" else 1):
        outs = []
        for l, a in zip(libs, algs):
            a.ctx.timing(reset=True)
            t0 = time.perf_counter()
            n = a.solve(xs[Bm:]); o = a.getSolution()
            t = time.perf_counter() - t0
            st = a.ctx.timing(reset=True)
            outs.append(o)
            print("%s %-22s rep %d: %8.1f terminated/s  %8.1f converged/s  ipm launch %.2f ms x %d  mean ipm iters %.2f" % (
                name, os.path.basename(l), rep, Bm / t, n / t, st["ms_socp"] / max(st["n_socp"], 1), st["n_socp"], o["ipm_iters"].mean()), flush=True)
    print(name, "bitwise equal:", {k: bool(np.array_equal(outs[0][k], outs[1][k])) for k in KEYS}, flush=True)
    for a in algs:
        a.ctx.close()
