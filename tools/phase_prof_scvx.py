"""In-kernel phase timing of ipm_kernel in SCvx mode (library built with -DIPM_PROFILE): cycles per phase of every instance's LAST
sub-problem solve of a whole scpp_hip_scvx_solve (the debug record of a launch overwrites the previous one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scpp_amd
lib = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = scpp_amd.RocketQuat().loadParameters()
alg = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=B, library=os.path.abspath(lib), max_iterations=int(sys.argv[3]) if len(sys.argv) > 3 else 3).initialize()
x0 = m.randomized_initial_states(B)
alg.solve(x0)
info = alg.ctx.socp_info()
names = ["init", "residuals", "scalings", "  factorFused(in 5)", "rhs(t,bx)+kktPrep", "sweeps+kktFinish", "dz/ds/step", "update", "prepareFactor", "  bwdSweeps(in 5)", "  fwdSweep(in 5)", "kernel_total"]
it = info[:, 4].mean()
p = info[:, 8:20].mean(axis=0)
print(f"SCvx B={B} mean ipm iters of the last solve {it:.1f}")
for n, v in zip(names, p):
    print(f"  {n:22s} total {v:14.0f}   per-iter {v / it:12.0f}   share {100 * v / p[11]:5.1f}%")
f = info[:, 20:30].mean(axis=0)
if f[3] > 0:
    print(f"  factor sweep detail (per call, {f[3]:.1f} calls): elimination<16> {f[0] / f[3]:.0f}  elimination<NL> {f[1] / f[3]:.0f}  whole sweep {f[2] / f[3]:.0f}"
          f"  -> eliminations {100 * (f[0] + f[1]) / f[2]:.1f}% of the sweep;"
          f" stage head (loads, H tile, Z'Z) {f[4] / f[3]:.0f}  between the eliminations {f[5] / f[3]:.0f}  tail {(f[2] - f[0] - f[1] - f[4] - f[5]) / f[3]:.0f}")
    print(f"  stage head split (per call): tail of the previous stage (store Ti, transpose, Z, forward pass) {f[6] / f[3]:.0f}  loads issued {f[7] / f[3]:.0f}"
          f"  H tile {f[8] / f[3]:.0f}  Z'Z + add {f[9] / f[3]:.0f}")
