"""Batched receding-horizon driver: host mirror of the reference's SC_sim executable
(scpp/src/SC_sim.cpp:19-104) for B independent closed loops (Monte-Carlo over initial states).

Per simulation step, exactly as the reference does for its single loop:
  solve(warm_start = step > 0)            SC_sim.cpp:45-46   (warm start: SCAlgorithm.cpp:141-145 -- the previous
                                                              solution is the initial trajectory, loadParameters() is
                                                              skipped so weight_trust_region_trajectory keeps its doubled
                                                              value, thrust_const is refreshed from the new trajectory)
  u0 = U[0], u1 = interpolatedInput(U, dt, t, FOH)            SC_sim.cpp:49-51, commonFunctions.cpp:6-19
  x <- simulate(model, dt, u0, u1, x)     SC_sim.cpp:53      (x aliases model.p.x_init: the plant state is the next
                                                              solve's initial-state constraint, SC_sim.cpp:36)
  stop when ||x - x_final|| < 0.02 or t < 0.25                SC_sim.cpp:57-62
Loops that stopped (or whose sub-problem failed: the reference would terminate) are masked out of later solves.
All numerical work runs in libscpp_hip.so (scpp_hip_sc_setup / sc_set_active / sc_solve / simulate)."""
import time

import numpy as np



def interpolated_input(U, t, total_time, first_order_hold=True):
    """commonFunctions.cpp:6-19 for a batch: U [B][K][nu], total_time [B] -> [B][nu]."""
    U = np.asarray(U, dtype=np.float64)
    B, K, _ = U.shape
    total_time = np.broadcast_to(np.asarray(total_time, dtype=np.float64), (B,))
    time_step = total_time / (K - 1)
    i = np.minimum((t / time_step).astype(np.int64), K - 2)
    rows = np.arange(B)
    u0 = U[rows, i]
    u1 = U[rows, i + 1] if first_order_hold else u0
    t_intermediate = np.fmod(t, time_step) / time_step
    return u0 + (u1 - u0) * t_intermediate[:, None]


class SCSim:
    def __init__(self, algorithm, time_step=0.05, max_steps=100):
        self.alg = algorithm
        self.time_step = float(time_step)
        self.max_steps = int(max_steps)

    def run(self, x_init):
        alg, ctx, model = self.alg, self.alg.ctx, self.alg.model
        x = np.array(x_init, dtype=np.float64).reshape(-1, 14)
        B = x.shape[0]
        x_final = np.array(list(model.p.x_final), dtype=np.float64)
        par_dim = np.tile(model.flow_params(nondimensionalize=False), (B, 1))  # dimensional flow-map parameters for the plant
        active = np.ones(B, dtype=np.int32)
        steps = np.zeros(B, dtype=np.int32)
        reached = np.zeros(B, dtype=bool)
        failed = np.zeros(B, dtype=bool)
        # per-step records of the whole batch + the mask of the loops that took the step (round 6: until then every loop's record was appended in a
        # Python loop per step, 57 ms per step for 4096 loops -- 11 of the 86 s of BASELINE configs[3] at size)
        rec_ok, rec_x, rec_u, rec_t, rec_it = [], [], [], [], []
        # what a step needs from the solve: the inputs, the planned time, the status and the iteration count -- not the 23 MB of planned states of
        # 4096 loops; the destination arrays are reused from step to step (`host_profile` in the result says where the host's share of a step goes)
        need = ("U", "sigma", "status", "sc_iters")
        out = None
        prof = dict(setup=0.0, solve=0.0, download=0.0, plant=0.0, bookkeeping=0.0)
        clock = time.perf_counter
        foh = bool(alg.opts.interpolate_input)
        for step in range(self.max_steps):
            if not active.any():
                break
            t0 = clock()
            ctx.sc_setup(model.p, alg.opts, x, warm_start=step > 0)
            ctx.sc_set_active(active)
            t1 = clock()
            ctx.sc_solve()
            t2 = clock()
            out = ctx.download(fields=need, out=out)
            t3 = clock()
            ok = (active != 0) & (out["status"] == 0)
            failed |= (active != 0) & (out["status"] != 0)
            u0 = out["U"][:, 0, :].copy()
            u1 = interpolated_input(out["U"], self.time_step, out["sigma"], foh)
            t4 = clock()
            ctx.set_flow_params(par_dim)
            x_new = ctx.simulate(self.time_step, u0, u1, x)
            t5 = clock()
            x = np.where(ok[:, None], x_new, x)
            rec_ok.append(ok)
            rec_x.append(x_new)
            rec_u.append(u0)
            rec_t.append(out["sigma"].copy())
            rec_it.append(out["sc_iters"].copy())
            steps += ok
            end = ok & ((np.linalg.norm(x - x_final, axis=1) < 0.02) | (out["sigma"] < 0.25))
            reached |= end
            active = (ok & ~end).astype(np.int32)
            t6 = clock()
            prof["setup"] += t1 - t0
            prof["solve"] += t2 - t1
            prof["download"] += t3 - t2
            prof["plant"] += t5 - t4
            prof["bookkeeping"] += (t4 - t3) + (t6 - t5)
        n = len(rec_ok)
        OK = np.array(rec_ok).reshape(n, B)
        XS, US = np.array(rec_x).reshape(n, B, 14), np.array(rec_u).reshape(n, B, 4)
        TS, IT = np.array(rec_t).reshape(n, B), np.array(rec_it, dtype=np.int32).reshape(n, B)
        return dict(
            X_sim=[XS[OK[:, b], b] for b in range(B)],
            U_sim=[US[OK[:, b], b] for b in range(B)],
            t_plan=[TS[OK[:, b], b] for b in range(B)],
            sc_iters=[IT[OK[:, b], b] for b in range(B)],
            steps=steps, reached_end=reached, solver_failed=failed, x=x, host_profile=prof,
        )
