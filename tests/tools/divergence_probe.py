"""Where do the device and the twin part ways on an instance whose decision records differ at a NON-tie?  Per iteration: relative distance of the two
iterates (states, inputs), both records.  Test infrastructure (uses the oracle).   usage: divergence_probe.py first inst [inst ...]   (GPU box)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scpp_amd, oracle_lib as O, scvx_audit
first = int(sys.argv[1]); insts = [int(a) for a in sys.argv[2:]]
m = scpp_amd.RocketQuat().loadParameters()
K, seed = 50, 20260927
x0 = np.concatenate([m.randomized_initial_states(1, first=first + b) for b in insts])
alg = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=len(insts)).initialize()
maxit = int(alg.opts.max_iterations)
path = scvx_audit.device_path(alg, x0, maxit)
out = {}
for i, b in enumerate(insts):
    s = O.SCvx(K=K); s.randomize(seed, first + b); s.set_solver(1); s.solve(); meta = s.meta(); info = s.info()
    # twin iterations: group the info rows (one per solve) by the decision code that ends an iteration
    its, cur = [], []
    for row in info:
        cur.append(row)
        if row[6] != 0:
            its.append(cur); cur = []
    rows = []
    nd = len(path)
    for j in range(1, min(nd, meta["n_all_td"])):
        Xt, Ut, _ = s.iterate(j)
        ms, rs = x0[i, 0], np.linalg.norm(x0[i, 1:4])  # nondimensionalize (rocketQuat.cpp:146-160): the oracle's all_td are nondimensional, the device's record is not
        sx = np.array([ms] + [rs] * 6 + [1.] * 7); su = np.array([ms * rs] * 3 + [ms * rs * rs])
        Xd, Ud = path[j]["X"][i] / sx, path[j]["U"][i] / su
        r = its[j - 1][-1] if j - 1 < len(its) else None
        rows.append(dict(iteration=j, rel_dX=float(np.abs(Xd - Xt).max() / np.abs(Xt).max()), rel_dU=float(np.abs(Ud - Ut).max() / np.abs(Ut).max()),
                         device_solves=int(path[j]["solves"][i]), device_radius=float(path[j]["radius"][i]),
                         twin_solves_in_iteration=len(its[j - 1]) if j - 1 < len(its) else None,
                         twin_rho=float(r[4]) if r is not None else None, twin_radius=float(r[5]) if r is not None else None, twin_J=float(r[1]) if r is not None else None,
                         twin_actual=float(r[2]) if r is not None else None, twin_predicted=float(r[3]) if r is not None else None))
        print("instance %d iteration %2d: rel dX %.1e rel dU %.1e | device solves %d radius %.3g | twin solves-in-it %s rho %s radius %s" % (
            b, j, rows[-1]["rel_dX"], rows[-1]["rel_dU"], rows[-1]["device_solves"], rows[-1]["device_radius"], rows[-1]["twin_solves_in_iteration"],
            "%.4f" % rows[-1]["twin_rho"] if r is not None else "-", "%.3g" % rows[-1]["twin_radius"] if r is not None else "-"), flush=True)
    out[str(b)] = dict(rows=rows, twin=dict(iterations=meta["iterations"], solves=meta["solves"], converged=meta["converged"]),
                       device=dict(iterations=int(path[-1]["iters"][i]), solves=int(path[-1]["solves"][i]), converged=int(path[-1]["converged"][i])))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(first=first, instances=out), open(os.path.join(ROOT, "gpurun_out", "r06_divergence_probe.json"), "w"), indent=1)
