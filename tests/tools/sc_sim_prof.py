"""Where the time of BASELINE configs[3] goes (SC_sim: B closed loops, warm-started SCAlgorithm solve + plant step per step): host wall time per call of the
Python mirror's loop and the library's kernel timers.  usage: sc_sim_prof.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
from scpp_amd.sc_sim import interpolated_input
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = scpp_amd.RocketQuat().loadParameters()
alg = scpp_amd.SCAlgorithm(m, K=50, batch_max=B).initialize()
ctx = alg.ctx
x = m.randomized_initial_states(B, first=90_000)
par_dim = np.tile(m.flow_params(nondimensionalize=False), (B, 1))
T = dict(setup=0., solve=0., download=0., simulate=0., host=0.)
active = np.ones(B, dtype=np.int32)
ctx.timing(reset=True)
t_all = time.time()
for step in range(steps):
    t0 = time.time(); ctx.sc_setup(m.p, alg.opts, x, warm_start=step > 0); ctx.sc_set_active(active); ctx.synchronize(); t1 = time.time()
    ctx.sc_solve(); ctx.synchronize(); t2 = time.time()
    out = ctx.download(); t3 = time.time()
    u0 = out["U"][:, 0, :]; u1 = interpolated_input(out["U"], 0.05, out["sigma"], True)
    t4 = time.time(); ctx.set_flow_params(par_dim); x = ctx.simulate(0.05, u0, u1, x); t5 = time.time()
    T["setup"] += t1 - t0; T["solve"] += t2 - t1; T["download"] += t3 - t2; T["host"] += t4 - t3; T["simulate"] += t5 - t4
    if step in (0, 1, steps - 1):
        print("step", step, "solve %.1f ms" % (1e3 * (t2 - t1)), "sc_iters mean %.2f" % out["sc_iters"].mean(), "ipm iters per solve %.1f" % (out["ipm_iters"].mean() / max(out["sc_iters"].mean(), 1)), ctx.timing(reset=True))
print("B=%d steps=%d total %.2f s" % (B, steps, time.time() - t_all), {k: round(v, 3) for k, v in T.items()})
