"""GPU trace of the first sub-problem of soak instance 454733 (the one failure in a million under split step lengths): the interior-point loop has no
device-side print, so the solve is repeated with maxit = 1 .. 30 and the final iterate's termination quantities are read back (scpp_hip_download_socp_info)
-- for the shipped library and for build/common_step.so.  usage (GPU box): python tools/r06_trace_failure.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scpp_amd

m = scpp_amd.RocketQuat().loadParameters()
x0 = m.randomized_initial_states(1, first=int(os.environ.get("INSTANCE", "454733")))
for lib in ("scpp_amd/libscpp_hip.so", "build/common_step.so"):
    path = os.path.join(ROOT, lib)
    if not os.path.exists(path):
        continue
    print("==", lib)
    for maxit in list(range(14, 31)) + [60]:
        a = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=1, library=path, max_iterations=1).initialize()
        a.ctx.set_socp_opts(1e-8, 1e-7, 1e-7, maxit)
        a.solve(x0)
        o = a.getSolution(); info = a.ctx.socp_info()[0]
        print("maxit %2d status %2d ipm %2d | pcost %.9e gap %.3e pres %.3e dres %.3e  info[6:10] %s" % (maxit, o["status"][0], o["ipm_iters"][0], info[0], info[1], info[2], info[3], np.array2string(info[6:10], precision=3)))
        a.ctx.close()
        if o["ipm_iters"][0] < maxit:  # the loop ended on its own (converged, or broke down): not the iteration limit's reduced-accuracy exit
            break
