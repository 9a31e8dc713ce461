"""Diagnostic: the literal audit rows of the shipped Rocket2D SCvx configuration at K = 30 on the GPU (tests/test_emu_kernels.py::_rocket2d_scvx_case)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scpp_amd, oracle_lib as oracle, scvx_audit
K = 30
m2 = scpp_amd.Rocket2D().loadParameters()
alg = scpp_amd.SCvxAlgorithm(m2, K=K, batch_max=4).initialize()
x0 = np.tile(m2.x_init, (2, 1)); x0[1:] = m2.randomized_initial_states(1, first=1)
alg.solve(x0); o = alg.getSolution()
print("device: iters", o["sc_iters"], "solves", o["solves"], "status", o["status"], "tr", o["trust_region"])
path = scvx_audit.device_path(alg, x0[:1], int(alg.opts.max_iterations))
for tol in (1e-9, 1e-8):
    s = oracle.SCvx(K=K, model=oracle.ROCKET2D, config_root=oracle.CONFIG_ROOT); s.set_tolerances(tol, tol, tol, 200)
    rows = scvx_audit.audit_rows(s, path, 0, alg.opts.alpha)
    print("literal tolerance", tol)
    for r in rows:
        gap = (r["cost"] - r["lit_cost"]) / abs(r["lit_cost"]) if r["lit_exitflag"] in (0, 10) else float("nan")
        print("  it %2d rej %d radius %.3e flag %3d cost %.12e lit %.12e gap %+.2e eq %.1e lp %.1e cone %.1e relU %s" % (
            r["iteration"], r["rejections"], r["radius"], r["lit_exitflag"], r["cost"], r["lit_cost"], gap, r["eq_violation"], r["min_lp_slack"], r["min_cone_slack"], r["relU"]))
