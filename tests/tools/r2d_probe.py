"""Debug probe: Rocket2D SC loop one iteration at a time (device or emulator library given as argv[1])."""
import numpy as np, scpp_amd, sys
from scpp_amd.sc_algorithm import load_sc_opts
m = scpp_amd.Rocket2D().loadParameters()
B = 2
alg = scpp_amd.SCAlgorithm(m, K=30, batch_max=B, library=(sys.argv[1] if len(sys.argv) > 1 else None)).initialize()
x0 = np.tile(m.x_init, (B, 1))
alg.opts = load_sc_opts(m.getParameterFolder(), alg.opts.K)
alg.ctx.sc_setup(m.sc_params(), alg.opts, x0, warm_start=False)
for it in range(8):
    n = alg.ctx.sc_iterate()
    o = alg.ctx.download()
    print(it, "active", n, "sc_iters", o["sc_iters"], "ipm", o["ipm_iters"], "nu %.3e" % o["nu_norm"][0], "sumdelta %.3e" % o["sum_delta"][0],
          "conv", o["converged"], "status", o["status"], "sigma %.9f" % o["sigma"][0])
    if n == 0:
        break
