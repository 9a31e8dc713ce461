"""Linear MPC throughput (Rocket2D, shipped MPC.info, 32768 controllers, cold start per solve) of two libraries, alternating: the shipped one (MPC_SPLIT_STEPS = 1:
primal and dual step lengths of their own) against build/mpc_common_step.so (-DMPC_SPLIT_STEPS=0).  usage (GPU box): python tools/r06_mpc_ab.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scpp_amd

m2 = scpp_amd.Rocket2D().loadParameters()
m2.p.constrain_initial_final = False
B = 32768
x = m2.randomized_initial_states(B)
libs = {"shipped": "scpp_amd/libscpp_hip.so", "mpc_common_step": "build/mpc_common_step.so"}
out = {k: [] for k in libs}
for rnd in range(3):
    for name, rel in libs.items():
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        a = scpp_amd.MPCAlgorithm(m2, batch_max=B, library=path).initialize()
        a.setInitialState(x); a.setFinalState(m2.p.x_final)
        a.solve()
        t0 = time.perf_counter()
        for _ in range(10):
            nok = a.solve()
        dt = (time.perf_counter() - t0) / 10
        o = a.getSolution()
        out[name].append({"solves_per_s": B / dt, "solved": int(nok), "mean_ipm_iterations": float(np.mean(o["iters"])), "mean_cost": float(np.mean(o["cost"]))})
        print(name, rnd, out[name][-1], flush=True)
        a.ctx.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_ab_mpc_split_steps.json"), "w"), indent=1)
