"""Interior-point warm start of consecutive SCvx sub-problems, studied in the scalar twin (oracle/structured_ipm.hpp) on the CPU:
iterations per solve by class (first solve / after an accepted step / after a rejected candidate) for
  mode 0  today's start: previous optimum, slacks and duals pushed theta into the cone
  mode 1  a CENTRED iterate of the previous solve (the first whose mu fell below snap_mu), unchanged
  mode 2  the previous optimum moved cone by cone onto mu-centred products (an active row keeps its multiplier and gets the slack
          mu / z, an inactive one keeps its slack and gets the multiplier mu / s); mu = snap_mu after an accepted step, mu_same for
          the re-solve after a rejection
The modes live in warm_start_study.patch (git apply it on oracle/, make -C oracle, run this, git checkout oracle/): they are not part
of the oracle.  Result (16 runs, ~400 solves, round 3): mode 0 16.0 iterations per solve after an accepted step / 9.4 after a
rejection; mode 1 22 .. 52 / 24 .. 42 (the data of consecutive sub-problems differ too much for an un-shifted interior iterate: the
residuals dwarf mu and the steps collapse); mode 2 mu = 0.3 / 1 / 3 / 10 / 30: 23.9 / 20.7 / 18.3 / 16.7 / 15.9 after an accepted
step, 11 .. 16 after a rejection.  A centred start converges cleanly (x10 per iteration) but has to begin at mu >= 10 to dominate
the residuals of the new linearisation, which costs what the centring saves.  Nothing here beats mode 0; not ported.
usage: python tools/experiments/warm_start_study.py [N instances]"""
import os, sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import ctypes as C

import oracle_lib as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = 50


def run(args):
    b, mode, snap_mu, theta, mu_same = args
    s = O.SCvx(K=K); s.randomize(20260927, 500_000 + b); s.set_solver(1)
    O.lib().oracle_scvx_set_twin_warm(s.h, int(mode), C.c_double(snap_mu), C.c_double(theta), C.c_double(mu_same))
    rc = s.solve()
    m = s.meta(); rows = s.info()
    X, U, _ = s.iterate(-1)
    return rc, m["iterations"], m["solves"], m["converged"], rows, X, U


def study(mode, snap_mu, theta, base=None, mu_same=1e-2):
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        res = list(ex.map(run, [(b, mode, snap_mu, theta, mu_same) for b in range(N)]))
    first, acc, rej = [], [], []
    for r in res:
        prev = None
        for row in r[4]:
            it = row[7]
            (first if prev is None else rej if prev == 0 else acc).append(it)
            prev = int(row[6])
    tot = sum(sum(row[7] for row in r[4]) for r in res)
    line = "mode %d mu_same %.0e snap_mu %.0e theta %.0e: conv %d/%d, scvx iters %.2f solves %.2f, ipm/traj %.1f | first %.1f, after accept %.2f (n=%d), after reject %.2f (n=%d), fails %d" % (
        mode, mu_same, snap_mu, theta, sum(r[3] for r in res), N, np.mean([r[1] for r in res]), np.mean([r[2] for r in res]), tot / N,
        np.mean(first), np.mean(acc), len(acc), np.mean(rej) if rej else 0, len(rej), sum(r[0] != 0 for r in res))
    if base is not None:
        same = [r[1] == q[1] and r[2] == q[2] for r, q in zip(res, base)]
        dX = [np.abs(r[5] - q[5]).max() / np.abs(q[5]).max() for r, q, sm in zip(res, base, same) if sm]
        line += " | same record %d/%d, dX median %.1e max %.1e" % (sum(same), N, np.median(dX) if dX else 0, max(dX) if dX else 0)
    print(line, flush=True)
    return res


if __name__ == "__main__":
    base = study(0, 1e-3, 1e-2)
    for snap_mu in (0.3, 1., 3., 10., 30.):
        study(2, snap_mu, 1e-2, base, mu_same=1e-2)
    for mu_same in (1e-1, 1e-3, 1e-4):
        study(2, 3., 1e-2, base, mu_same=mu_same)
