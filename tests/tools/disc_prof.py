"""In-kernel phase timing of discretize_kernel (library built with -DDISC_PROFILE): one launch at B instances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
lib = os.path.abspath(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
m = scpp_amd.RocketQuat().loadParameters()
alg = scpp_amd.SCAlgorithm(m, K=50, batch_max=B, library=lib).initialize()
alg.ctx.sc_setup(m.p, alg.opts, m.randomized_initial_states(B))
alg.ctx.discretize(); alg.ctx.synchronize()
alg.ctx.timing(reset=True)
alg.ctx.discretize(); alg.ctx.synchronize()
print(alg.ctx.timing())
