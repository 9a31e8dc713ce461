"""Exploration for the parity bars: for every instance whose device record differs from the twin's, find the FIRST decision of
SCvxAlgorithm::iterate at which the two runs part ways and print how close the twin's decision variable was to its threshold."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scpp_amd, oracle_lib as oracle, scvx_audit
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K, first, seed = 50, 300_000, 20260927
model = scpp_amd.RocketQuat().loadParameters()
x0 = model.randomized_initial_states(N, first=first)
alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=N).initialize()
alg.solve(x0); out = alg.getSolution()
def twin(b):
    s = oracle.SCvx(K=K); s.randomize(seed, first + b); s.set_solver(1)
    rc = s.solve(); m = s.meta()
    return rc, m["iterations"], m["solves"], m["converged"], s.info()
with ThreadPoolExecutor(32) as ex:
    ref = list(ex.map(twin, range(N)))
same = np.array([r[0] == 0 and out["sc_iters"][b] == r[1] and out["solves"][b] == r[2] and out["converged"][b] == r[3] for b, r in enumerate(ref)])
bad = [int(b) for b in np.nonzero(~same)[0]]
print("non-identical:", bad)
o = alg.opts
path = scvx_audit.device_path(alg, x0[bad], int(o.max_iterations))
for i, b in enumerate(bad):
    info = ref[b][4]
    # twin per iteration: rows until an accepted one (code != 0)
    its, cur = [], []
    for row in info:
        cur.append(row)
        if row[6] != 0:
            its.append(cur); cur = []
    prev_s = 0
    for j, st in enumerate(path[1:]):
        if st["iters"][i] <= j:
            break
        nd = int(st["solves"][i] - prev_s); prev_s = int(st["solves"][i])
        if j >= len(its):
            print(b, "iteration", j + 1, "twin ended earlier (converged)", "last twin row pred", its[-1][-1][3]); break
        rows = its[j]; nt = len(rows)
        rd, rt = float(st["radius"][i]), float(rows[-1][5])
        conv_d = bool(st["converged"][i]) and st["iters"][i] == j + 1
        conv_t = rows[-1][6] == 3
        if nd != nt or abs(rd - rt) > 1e-12 * rt or conv_d != conv_t:
            r = rows[min(nd, nt) - 1]
            J = r[1]
            print("inst %d it %d: dev solves %d radius %.6g conv %d | twin solves %d radius %.6g conv %d | twin row at divergence: J %.3e actual %.3e pred %.3e rho %.3e code %d  -> |actual|/J %.1e, |rho-rho1| %.1e |rho-rho2| %.1e, ||pred|-thr|/thr %.1e"
                  % (b, j + 1, nd, rd, conv_d, nt, rt, conv_t, J, r[2], r[3], r[4], int(r[6]), abs(r[2]) / max(J, 1e-300), abs(r[4] - o.rho_1), abs(r[4] - o.rho_2), abs(abs(r[3]) - o.change_threshold) / o.change_threshold))
            break
