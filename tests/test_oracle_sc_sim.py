"""Oracle restatement of the SC_sim receding-horizon driver (oracle/sc_sim.hpp) and the host mirror's
interpolated_input (commonFunctions.cpp:6-19)."""
import numpy as np

import scpp_amd


def test_interpolated_input_matches_scalar_definition():
    rng = np.random.default_rng(3)
    B, K, nu = 5, 9, 4
    U = rng.normal(size=(B, K, nu))
    total = rng.uniform(1.0, 5.0, size=B)
    for t in (0.0, 0.05, 0.37, 0.999):
        got = scpp_amd.interpolated_input(U, t, total, True)
        for b in range(B):
            dt = total[b] / (K - 1)
            i = min(int(t / dt), K - 2)
            w = np.fmod(t, dt) / dt
            assert np.allclose(got[b], U[b, i] + (U[b, i + 1] - U[b, i]) * w, rtol=0, atol=1e-15)
        got0 = scpp_amd.interpolated_input(U, t, total, False)  # zero-order hold: u1 = u0
        for b in range(B):
            i = min(int(t / (total[b] / (K - 1))), K - 2)
            assert np.array_equal(got0[b], U[b, i])


def test_oracle_sc_sim_closed_loop_semantics(oracle):
    """x aliases x_init (SC_sim.cpp:36): after the loop the handle's x_init is the last plant state; the plant
    burns mass; every solve after the first is a warm start from the previous plan (plans stay close)."""
    K, steps = 8, 3
    sc = oracle.SC(oracle.ROCKETQUAT, K=K)
    sc.randomize(20260927, 1)
    sc.set_solver(1)
    x0 = sc.x_init().copy()
    r = sc.sim(0.05, steps)
    assert r["steps"] == steps and not r["solver_failed"]
    assert np.array_equal(sc.x_init(), r["X_sim"][-1])
    m = np.concatenate([[x0[0]], r["X_sim"][:, 0]])
    assert np.all(np.diff(m) < 0.0)
    # consecutive plans differ by roughly the elapsed time
    assert np.all(np.abs(np.diff(r["t_plan"])) < 0.5)
    # first applied input is the first node of the plan: thrust within bounds
    T = np.linalg.norm(r["U_sim"][:, :3], axis=1)
    assert np.all(T > 0.0)
