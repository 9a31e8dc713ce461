"""GPU probe for the linear MPC path (run on the GPU box): solve-level and closed-loop parity against the oracle twin, then
rates.  Test infrastructure (uses the oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scpp_amd, oracle_lib as O
m = scpp_amd.Rocket2D().loadParameters(); m.p.constrain_initial_final = False
BM = 32768
a = scpp_amd.MPCAlgorithm(m, batch_max=BM).initialize()
o = O.MPC()
B = 256
x0 = m.randomized_initial_states(B)
# a few states further down the descent, some of them outside their own constraints
x0[200:, 1] *= 0.3; x0[230:, 4] = 1.2; x0[240:, 0] = 2.0 * x0[240:, 1]
a.setInitialState(x0); a.setFinalState(m.p.x_final)
n = a.solve(); out = a.getSolution()
bad = 0; wU = 0.0; wX = 0.0; same_it = 0
for b in range(B):
    r = o.solve(x0[b], kind=1)
    if (r["status"] < 0) != (out["status"][b] < 0) or (r["status"] >= 0 and r["status"] != out["status"][b]):
        bad += 1; continue
    if r["status"] >= 0:
        wU = max(wU, np.abs(out["U"][b] - r["U"]).max() / np.abs(r["U"]).max()); wX = max(wX, np.abs(out["X"][b] - r["X"]).max())
        same_it += int(out["iters"][b] == r["iters"])
print(f"solve parity: {B} states, {n} solved, status mismatches {bad}, same iteration count {same_it}/{n}, worst rel dU {wU:.2e}, worst dX {wX:.2e}")
NS = 8
xs = m.randomized_initial_states(NS)
t = time.time(); r = scpp_amd.MPCSim(a).run(xs); td = time.time() - t
ws = 0; wx = 0.
for b in range(NS):
    q = o.sim(xs[b])
    ws += int(q["steps"] == r["steps"][b] and q["failed_solves"] == r["failed_solves"][b])
    wx = max(wx, np.abs(q["x"] - r["x"][b]).max())
print(f"closed-loop parity: {NS} loops x {r['steps'][0]} steps in {td:.2f}s: identical step/failure counts {ws}/{NS}, worst dx {wx:.2e}, failed solves {r['failed_solves'].tolist()}")
for Bt in (1024, 8192, 32768):
    xb = m.randomized_initial_states(Bt)
    a.setInitialState(xb); a.solve()
    a.ctx.timing(reset=True)
    t = time.time()
    for _ in range(5):
        a.solve()
    dt = (time.time() - t) / 5
    tm = a.ctx.timing(reset=True)
    st = a.getSolution()
    print(f"B={Bt}: {Bt/dt:.0f} solves/s wall (incl. upload + status download), kernel {tm['ms_socp']/tm['n_socp']:.3f} ms -> {Bt/(tm['ms_socp']/tm['n_socp'])*1e3:.0f} solves/s, mean IPM iterations {st['iters'].mean():.1f}, ok {int((st['status']>=0).sum())}")
Bs = 4096
xb = m.randomized_initial_states(Bs)
t = time.time(); r = scpp_amd.MPCSim(a, max_steps=300).run(xb); dt = time.time() - t
print(f"closed loops: B={Bs} x 300 steps in {dt:.2f}s = {Bs*300/dt:.0f} controller steps/s; failed solves total {int(r['failed_solves'].sum())}")
