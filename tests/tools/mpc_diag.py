"""Per-instance comparison of an MPC kernel build against the twin (GPU box).  usage: mpc_diag.py lib.so [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scpp_amd, oracle_lib as O
m = scpp_amd.Rocket2D().loadParameters(); m.p.constrain_initial_final = False
o = O.MPC()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x0 = m.randomized_initial_states(n)
a = scpp_amd.MPCAlgorithm(m, batch_max=n, library=os.path.join(ROOT, sys.argv[1])).initialize()
a.setInitialState(x0); a.setFinalState(m.p.x_final); a.solve(); out = a.getSolution()
hist = {}
for b in range(n):
    r = o.solve(x0[b], kind=1)
    dU = np.abs(out["U"][b] - r["U"]).max() / np.abs(r["U"]).max() if r["status"] >= 0 and out["status"][b] >= 0 else float('nan')
    key = (int(out["status"][b]), int(r["status"]), int(out["iters"][b]) - int(r["iters"]))
    hist[key] = hist.get(key, 0) + 1
    if key != (0, 0, 0) or dU > 1e-9:
        print(b, "dev", out["status"][b], out["iters"][b], "twin", r["status"], r["iters"], "dU", dU, "cost", out["cost"][b], r["input_cost"], r["error_cost"])
print("histogram (dev status, twin status, iteration difference):", hist)
