"""Primal and dual step lengths of their own (csrc/ipm_solve.h: IPM_SPLIT_STEPS, oracle/structured_ipm.hpp: RQSocpSettings::split_steps), studied in
the scalar twin on the CPU BEFORE it was built into the kernel (round 6): interior-point iterations per RocketQuat K = 50 SCvx trajectory and per solve
by class (first solve / after an accepted step / after a rejected candidate), ECOS's common step length against split ones.

Result (16 trajectories, seed 20260927, instances 500000 ..): common 366.1 iterations per trajectory (19.6 / 16.02 / 9.43 per solve), split 326.6
(17.5 / 14.21 / 8.87): -10.8 %, 16 / 16 converged either way, no solver failure.  Variants tried in a scratch copy of the twin and NOT kept: the centring
parameter from the mean / geometric mean / larger of the two affine step lengths instead of the common one (319.6 / 317.2 / 338.9: inside the spread
of the decision sequences, the per-solve counts are 14.17 / 14.14 / 14.88 after an accepted step); splitting only when the smaller step length is at
least 0.3 / 0.5 / 0.7 (15.72 / 15.73 / 15.88: the gain IS the iterations in which one of the two directions is blocked early); a cap of 2 / 4 / 10 on
the ratio of the two (14.81 / 14.40 / 14.23).
usage: python tools/experiments/split_step_study.py [N instances]"""
import os, sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle_lib as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = 50


def run(args):
    b, split = args
    s = O.SCvx(K=K); s.randomize(20260927, 500_000 + b); s.set_solver(1)
    O.lib().oracle_scvx_set_twin_split_steps(s.h, int(split))
    rc = s.solve()
    m = s.meta()
    return rc, m["iterations"], m["solves"], m["converged"], s.info()


for split in (0, 1):
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(run, [(b, split) for b in range(N)]))
    first, acc, rej = [], [], []
    for r in res:
        prev = None
        for row in r[4]:
            (first if prev is None else rej if prev == 0 else acc).append(row[7])
            prev = int(row[6])
    tot = sum(sum(row[7] for row in r[4]) for r in res)
    print("split_steps %d: converged %d / %d, SCvx iterations %.2f, solves %.2f, interior-point iterations per trajectory %.1f | per solve: first %.1f, "
          "after an accepted step %.2f, after a rejection %.2f; solver failures %d"
          % (split, sum(r[3] for r in res), N, np.mean([r[1] for r in res]), np.mean([r[2] for r in res]), tot / N, np.mean(first), np.mean(acc),
             np.mean(rej) if rej else 0.0, sum(r[0] != 0 for r in res)), flush=True)
