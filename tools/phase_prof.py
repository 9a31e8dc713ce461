"""In-kernel phase timing of ipm_kernel (library built with -DIPM_PROFILE): cycles per phase, one SC iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scpp_amd
lib = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
NIT = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # profile the NIT-th SC iteration (>= 2: warm-started solve)
m = scpp_amd.RocketQuat().loadParameters()
alg = scpp_amd.SCAlgorithm(m, K=50, batch_max=B, library=os.path.abspath(lib)).initialize()
x0 = m.randomized_initial_states(B)
alg.ctx.sc_setup(m.p, alg.opts, x0)
for _ in range(NIT - 1):
    alg.ctx.sc_iterate()
prev = alg.ctx.download()["ipm_iters"].copy() if NIT > 1 else 0
alg.ctx.timing(reset=True)
alg.ctx.sc_iterate()
info = alg.ctx.socp_info()
names = ["init", "residuals", "scalings", "  factorFused(in 5)", "rhs(t,bx)+kktPrep", "sweeps+kktFinish", "dz/ds/step", "update", "prepareFactor", "  bwdSweeps(in 5)", "  fwdSweep(in 5)", "kernel_total"]
it = info[:, 4].mean()
p = info[:, 8:20].mean(axis=0)
print(f"B={B} mean ipm iters {it:.1f}; counts are clock64() ticks = s_memtime, the shader clock (4096 instances = two generations of 2048 resident wavefronts in the 45.2 ms launch: 39.9 M ticks per wavefront lifetime of ~22.6 ms, i.e. ~1.77 GHz under this load)")
for n, v in zip(names, p):
    if n != "-":
        print(f"  {n:14s} total {v:14.0f}   per-iter {v / it:12.0f}   share {100 * v / p[11]:5.1f}%")
f = info[:, 20:26].mean(axis=0)
if f[3] > 0:
    print(f"  factor sweep detail (per call, {f[3]:.1f} calls): elimination<16> {f[0] / f[3]:.0f}  elimination<NL> {f[1] / f[3]:.0f}  whole sweep {f[2] / f[3]:.0f}"
          f"  -> eliminations {100 * (f[0] + f[1]) / f[2]:.1f}% of the sweep;"
          f" stage head (loads, H tile, Z'Z) {f[4] / f[3]:.0f}  between the eliminations {f[5] / f[3]:.0f}  tail {(f[2] - f[0] - f[1] - f[4] - f[5]) / f[3]:.0f}")
print(alg.ctx.timing())
