"""Device vs structured twin at scale (part (a) of test_scvx_at_scale_parity_and_literal_audit), for experiments with the twin's build.
usage: twin_scale.py [N]   (SCPP_ORACLE_LIBRARY selects the oracle build; prints the record / state / input statistics)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scpp_amd, oracle_lib as oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K, first, seed = 50, 300_000, 20260927
model = scpp_amd.RocketQuat().loadParameters()
x0 = model.randomized_initial_states(N, first=first)
alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=N).initialize()
nconv = alg.solve(x0); out = alg.getSolution()
def twin(b):
    s = oracle.SCvx(K=K); s.randomize(seed, first + b); s.set_solver(1)
    rc = s.solve(); m = s.meta(); X, U, _ = s.iterate(-1)
    return rc, m["iterations"], m["solves"], m["converged"], X, U
t0 = time.time()
threads = min(32, os.cpu_count() or 1)
with ThreadPoolExecutor(threads) as ex:
    ref = list(ex.map(twin, range(N)))
same = np.array([r[0] == 0 and out["sc_iters"][b] == r[1] and out["solves"][b] == r[2] and out["converged"][b] == r[3] for b, r in enumerate(ref)])
relX = np.array([np.abs(out["X"][b] - r[4]).max() / np.abs(r[4]).max() for b, r in enumerate(ref)])
relU = np.array([np.abs(out["U"][b] - r[5]).max() / np.abs(r[5]).max() for b, r in enumerate(ref)])
print("oracle", os.environ.get("SCPP_ORACLE_LIBRARY", "default"), "N", N, "identical records", int(same.sum()), "converged dev", nconv, "twin", sum(r[3] for r in ref),
      "| over identical: relX median %.1e p99 %.1e max %.1e, > 1e-5: %d | relU median %.1e p99 %.1e max %.1e, > 1e-5: %d | twin %.0f s"
      % (np.median(relX[same]), np.percentile(relX[same], 99), relX[same].max(), int((relX[same] > 1e-5).sum()),
         np.median(relU[same]), np.percentile(relU[same], 99), relU[same].max(), int((relU[same] > 1e-5).sum()), time.time() - t0))
