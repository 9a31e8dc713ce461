import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scpp_amd, oracle_lib as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = scpp_amd.RocketQuat().loadParameters()
a = scpp_amd.SCvxAlgorithm(m, batch_max=B).initialize()
x0 = m.randomized_initial_states(B, first=FIRST)
n = a.solve(x0); out = a.getSolution()
bad = np.nonzero(out['status'] != 0)[0]
print('converged', n, 'failures', len(bad), 'status values', np.unique(out['status'], return_counts=True))
info = a.ctx.socp_info()
np.set_printoptions(linewidth=200, precision=4)
for b in bad[:6]:
    print('inst', b, 'status', out['status'][b], 'scvx iter', out['sc_iters'][b], 'solves', out['solves'][b], 'tr', out['trust_region'][b], 'last ipm info', info[b][:8])
    s = O.SCvx(K=50); s.randomize(20260927, int(b) + FIRST); s.set_solver(1); rc = s.solve(); mm = s.meta()
    print('   oracle rc', rc, 'iters', mm['iterations'], 'solves', mm['solves'], 'conv', mm['converged'])
