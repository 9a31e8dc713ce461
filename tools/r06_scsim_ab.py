"""SC_sim-shaped work (warm-started SCAlgorithm solves, BASELINE configs[3]) under split and common step lengths: 4096 closed loops x 12 steps on the shipped
library and on build/common_step.so (-DIPM_SPLIT_STEPS=0), alternating; per step the wall time of scpp_hip_sc_solve and the interior-point iterations per
solve (cold first step apart from the warm-started ones).  usage (GPU box): python tools/r06_scsim_ab.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scpp_amd
from scpp_amd.sc_sim import interpolated_input

m = scpp_amd.RocketQuat().loadParameters()
B, steps = 4096, 12
x00 = m.randomized_initial_states(B, first=90_000)
par_dim = np.tile(m.flow_params(nondimensionalize=False), (B, 1))
out = {}
for rnd in range(2):
    for name, rel in (("shipped", "scpp_amd/libscpp_hip.so"), ("common_step", "build/common_step.so")):
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        a = scpp_amd.SCAlgorithm(m, K=50, batch_max=B, library=path).initialize()
        x = x00.copy(); rec = []
        for step in range(steps):
            a.ctx.sc_setup(m.p, a.opts, x, warm_start=step > 0)
            t0 = time.perf_counter(); a.ctx.sc_solve(); dt = time.perf_counter() - t0
            o = a.ctx.download(fields=("U", "sigma", "status", "ipm_iters", "sc_iters"))
            rec.append((dt, float(o["ipm_iters"].mean()), float(o["sc_iters"].mean()), int((o["status"] != 0).sum())))
            u0 = o["U"][:, 0, :].copy(); u1 = interpolated_input(o["U"], 0.05, o["sigma"], True)
            a.ctx.set_flow_params(par_dim)
            x = a.ctx.simulate(0.05, u0, u1, x)
        a.ctx.close()
        warm = rec[2:]
        out.setdefault(name, []).append({"cold_solve_s": rec[0][0], "cold_ipm_iterations": rec[0][1], "warm_solve_s_mean": float(np.mean([r[0] for r in warm])),
                                         "warm_ipm_iterations_per_solve": float(np.mean([r[1] for r in warm])), "sc_iterations": float(np.mean([r[2] for r in warm])),
                                         "failures": int(sum(r[3] for r in rec))})
        print(name, rnd, out[name][-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_ab_scsim_split_steps.json"), "w"), indent=1)
