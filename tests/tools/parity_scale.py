"""Parity at scale (run on the GPU box): N instances of the headline workload, device vs oracle (structured twin,
multi-threaded), reporting the worst relative deviations.  Test infrastructure (uses the oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scpp_amd, oracle_lib as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 48
m = scpp_amd.RocketQuat().loadParameters()
a = scpp_amd.SCAlgorithm(m, K=50, batch_max=N).initialize()
x0 = m.randomized_initial_states(N)
t = time.time(); a.solve(x0); out = a.getSolution(); tg = time.time() - t
t = time.time(); ref = O.sc_batch(50, 20260927, 0, N, nthreads=min(32, os.cpu_count() or 1), solver=1); tc = time.time() - t
relX = np.abs(out["X"] - ref["X"]).max(axis=(1, 2)) / np.abs(ref["X"]).max(axis=(1, 2))
relU = np.abs(out["U"] - ref["U"]).max(axis=(1, 2)) / np.abs(ref["U"]).max(axis=(1, 2))
print(f"SC mode, {N} instances: device {tg:.2f}s, oracle {tc:.1f}s; worst rel dX {relX.max():.2e}, worst rel dU {relU.max():.2e}, "
      f"worst rel dsigma {(np.abs(out['sigma'] - ref['t']) / ref['t']).max():.2e}; SC iterations equal: {bool((out['sc_iters'] == ref['iters']).all())}; "
      f"IPM iteration totals equal for {int((out['ipm_iters'] == ref['ipm_iters']).sum())}/{N}; device failures {int((out['status'] != 0).sum())}")
a.ctx.close()
v = scpp_amd.SCvxAlgorithm(m, batch_max=NV).initialize()
xv = m.randomized_initial_states(NV)
v.solve(xv); vo = v.getSolution()
wX = 0.0; same = 0
for b in range(NV):
    s = O.SCvx(K=50); s.randomize(20260927, b); s.set_solver(1); rc = s.solve(); mm = s.meta()
    X, U, tt = s.iterate(-1)
    wX = max(wX, np.abs(vo["X"][b] - X).max() / np.abs(X).max())
    same += int(vo["sc_iters"][b] == mm["iterations"] and vo["solves"][b] == mm["solves"] and vo["converged"][b] == mm["converged"])
print(f"SCvx mode, {NV} instances: identical iteration / solve / convergence record for {same}/{NV}; worst rel dX {wX:.2e}")
