"""MPC leg of bench.py on its own (for rocprofv3): solve rate at 32768 controllers and 300-step closed loops of 4096."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
m = scpp_amd.Rocket2D().loadParameters(); m.p.constrain_initial_final = False
B = 32768
a = scpp_amd.MPCAlgorithm(m, batch_max=B).initialize()
x = m.randomized_initial_states(B)
a.setInitialState(x); a.setFinalState(m.p.x_final); a.solve(); a.ctx.timing(reset=True)
t = time.perf_counter()
for _ in range(10):
    n = a.solve()
dt = (time.perf_counter() - t) / 10
tm = a.ctx.timing(reset=True)
print(f"mpc_solve_kernel: B={B} avg launch {tm['ms_socp']/tm['n_socp']:.3f} ms (HIP events), {B/dt:.0f} solves/s wall, solved {n}")
t = time.perf_counter(); r = scpp_amd.MPCSim(a, max_steps=300).run(x[:4096]); dt = time.perf_counter() - t
print(f"closed loops: 4096 x 300 steps in {dt:.3f}s = {r['steps'].sum()/dt:.0f} controller steps/s")
