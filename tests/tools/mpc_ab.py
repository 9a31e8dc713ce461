"""A/B of MPC kernel builds on the GPU box: parity against the twin on 64 states + solve rate.  usage: mpc_ab.py lib.so ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scpp_amd, oracle_lib as O
m = scpp_amd.Rocket2D().loadParameters(); m.p.constrain_initial_final = False
o = O.MPC()
x0 = m.randomized_initial_states(64)
ref = [o.solve(x0[b], kind=1) for b in range(64)]
for lib in sys.argv[1:]:
    a = scpp_amd.MPCAlgorithm(m, batch_max=32768, library=os.path.join(ROOT, lib)).initialize()
    a.setInitialState(x0); a.setFinalState(m.p.x_final); a.solve(); out = a.getSolution()
    bad = sum(int(out["status"][b] != ref[b]["status"] or out["iters"][b] != ref[b]["iters"]) for b in range(64))
    wU = max(np.abs(out["U"][b] - ref[b]["U"]).max() / np.abs(ref[b]["U"]).max() for b in range(64))
    xb = m.randomized_initial_states(32768)
    a.setInitialState(xb); a.solve(); a.ctx.timing(reset=True)
    for _ in range(5):
        a.solve()
    tm = a.ctx.timing(reset=True)
    ms = tm['ms_socp'] / tm['n_socp']
    print(f"{lib}: status/iteration mismatches {bad}/64, worst rel dU {wU:.1e}; B=32768 kernel {ms:.3f} ms -> {32768/ms*1e3:.0f} solves/s", flush=True)
    a.ctx.close()
