"""Dumps the results of fixed SC / SCvx runs of the library of the tree this script is started in (A/B of two source trees on one GPU box).
usage (from a tree's root): python <path>/dump_runs.py out.npz"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scpp_amd
out = {}
m = scpp_amd.RocketQuat().loadParameters()
x0 = m.randomized_initial_states(256)
a = scpp_amd.SCAlgorithm(m, K=50, batch_max=256).initialize(); a.solve(x0); o = a.getSolution(); a.ctx.close()
for k in ("X", "U", "sigma", "ipm_iters", "status", "sc_iters"): out["sc_" + k] = o[k]
v = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=256).initialize(); v.solve(x0); o = v.getSolution(); v.ctx.close()
for k in ("X", "U", "sigma", "ipm_iters", "status", "sc_iters", "solves"): out["scvx_" + k] = o[k]
m2 = scpp_amd.Rocket2D().loadParameters()
x2 = np.tile(m2.x_init, (2, 1)); x2[1:] = m2.randomized_initial_states(1, first=1)
for K in (30,):
    v = scpp_amd.SCvxAlgorithm(m2, K=K, batch_max=4).initialize(); v.solve(x2); o = v.getSolution(); v.ctx.close()
    for k in ("X", "U", "sigma", "ipm_iters", "status", "sc_iters", "solves", "trust_region"): out["r2d_scvx_" + k] = o[k]
    a = scpp_amd.SCAlgorithm(m2, K=K, batch_max=4).initialize(); a.solve(x2); o = a.getSolution(); a.ctx.close()
    for k in ("X", "U", "sigma", "ipm_iters", "status", "sc_iters"): out["r2d_sc_" + k] = o[k]
np.savez(sys.argv[1], **out)
print("dumped", sys.argv[1], {k: v.shape for k, v in out.items() if k.endswith("_X")})
