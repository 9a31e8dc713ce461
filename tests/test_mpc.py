"""Linear MPC path of the product: host setup (exactLinearDiscretization + elimination of the
states, csrc/mpc_setup.h), the one-wavefront-per-controller solve kernel and the closed-loop driver (csrc/mpc_kernel.h),
through the C ABI.  CPU tests run the kernel SOURCES on the wave emulator; the `gpu` tests run the HIP build.  The checker is
oracle/mpc.hpp (condensed twin pinned in tests/test_oracle_mpc.py)."""
import ctypes as C

import numpy as np
import pytest

import scpp_amd
from scpp_amd import _lib

E_ARG, E_HIP, E_UNSUPPORTED, E_STATE = -1, -2, -3, -4


@pytest.fixture(scope="module")
def rocket2d():
    m = scpp_amd.Rocket2D().loadParameters()
    m.p.constrain_initial_final = False  # model.info: "enable for SC and disable for MPC/LQR"
    return m


def _alg(rocket2d, library, batch_max=16, **kw):
    return scpp_amd.MPCAlgorithm(rocket2d, batch_max=batch_max, library=library).initialize(**kw)


def _check_solves(oracle, alg, x0, rtolU=1e-7, atolX=1e-6):
    """Same status and iteration count as the twin; plan and costs to 1e-7 relative -- one order above the interior-point
    termination tolerance (1e-8): a plan whose thrust levels sit strictly inside their bounds is determined only to about
    that accuracy, and kernel and twin round differently (shared reciprocals, FMA contraction, summation trees)."""
    o = oracle.MPC()
    alg.setInitialState(x0); alg.setFinalState(alg.model.p.x_final)
    n = alg.solve()
    out = alg.getSolution()
    assert n == int((out["status"] >= 0).sum())
    for b in range(x0.shape[0]):
        r = o.solve(x0[b], kind=1)
        assert out["status"][b] == r["status"]
        if r["status"] >= 0:
            assert out["iters"][b] == r["iters"]
            assert np.abs(out["U"][b] - r["U"]).max() <= rtolU * np.abs(r["U"]).max()
            assert np.abs(out["X"][b] - r["X"]).max() <= atolX
            assert abs(out["cost"][b][0] - r["input_cost"]) <= rtolU * r["input_cost"]
            assert abs(out["cost"][b][1] - r["error_cost"]) <= rtolU * r["error_cost"]
    return out


def test_shipped_configuration_is_guarded(emu_lib):
    m = scpp_amd.Rocket2D().loadParameters()
    assert m.p.constrain_initial_final  # as shipped (for SC)
    with pytest.raises(RuntimeError):
        scpp_amd.MPCAlgorithm(m, library=emu_lib).initialize()


def test_exact_linear_discretization_matches_oracle(oracle, rocket2d, emu_lib):
    a = _alg(rocket2d, emu_lib)
    o = oracle.MPC()
    assert a.K == o.K == 7
    assert np.abs(a.A - o.A).max() < 1e-14 and np.abs(a.B - o.B).max() < 1e-14 and np.abs(a.z - o.z).max() < 1e-13
    a.ctx.close()


def test_emu_mpc_solve_matches_twin(oracle, rocket2d, emu_lib):
    a = _alg(rocket2d, emu_lib)
    x0 = rocket2d.randomized_initial_states(5)
    x0[0] = rocket2d.p.x_init
    x0[3, 1] *= 0.4                       # further down the descent
    x0[4, 4] = 1.3                        # tilt outside its box: the reference problem is infeasible at k = 0
    out = _check_solves(oracle, a, x0)
    assert out["status"][4] == -3 and (out["status"][:4] == 0).all()
    # a failed solve leaves the previous solution of that controller untouched
    prev = a.getSolution()
    x1 = x0.copy(); x1[0, 5] = 1.0        # rate outside its box
    a.setInitialState(x1); a.solve()
    now = a.getSolution()
    assert now["status"][0] == -3 and np.array_equal(now["U"][0], prev["U"][0]) and np.array_equal(now["X"][0], prev["X"][0])
    a.ctx.close()


def test_emu_mpc_closed_loop_matches_oracle(oracle, rocket2d, emu_lib):
    """MPC_sim.cpp:49-86 on the emulated device: solve -> plant step under the previous input -> apply U[0]."""
    a = _alg(rocket2d, emu_lib)
    o = oracle.MPC()
    x0 = rocket2d.randomized_initial_states(3, first=11)
    r = scpp_amd.MPCSim(a, max_steps=12).run(x0)
    for b in range(3):
        q = o.sim(x0[b], max_steps=12)
        assert r["steps"][b] == q["steps"] == 12 and r["failed_solves"][b] == q["failed_solves"] and r["ipm_iters"][b] == q["ipm_iters"]
        assert np.abs(r["x"][b] - q["x"]).max() < 1e-9 * np.abs(q["x"]).max() and np.abs(r["u"][b] - q["u"]).max() <= 1e-8 * 420000.0
        assert abs(r["t"][b] - 0.12) < 1e-12
    # stop rule: a loop that starts at the target retires after its first step; the clock limit retires the others
    xs = np.vstack([rocket2d.p.x_final + np.array([0, 0.001, 0, 0, 0, 0]), x0[0]])
    r = scpp_amd.MPCSim(a, sim_time=0.03, stop_tol=0.5).run(xs)
    assert r["reached"].tolist() == [1, 0] and r["n_reached"] == 1
    assert r["steps"].tolist() == [1, o.sim(xs[1], sim_time=0.03)["steps"]]
    a.ctx.close()


def test_mpc_abi_errors(rocket2d, emu_lib):
    raw = _lib.load_library(emu_lib)
    a = _alg(rocket2d, emu_lib)           # a valid options block to start from
    h = C.c_void_p()
    assert raw.scpp_hip_create(C.byref(h), 0, scpp_amd.MODEL_ROCKET2D, 7, 4, 0) == 0
    x = np.zeros((4, 6)); xp = x.ctypes.data_as(C.c_void_p)
    n = C.c_int(0)
    assert raw.scpp_hip_mpc_solve(h, xp, xp, 4, C.byref(n)) == E_STATE          # before mpc_setup
    assert raw.scpp_hip_mpc_download(h, None, None, None, None, None) == E_STATE
    par = rocket2d.flow_params(); pp = par.ctypes.data_as(C.c_void_p)

    def opts(**kw):
        o = scpp_amd.MpcOpts()
        o.K, o.constant_dynamics, o.time_horizon = 7, 1, 1.5
        o.state_weights_terminal[:] = [5, 5, 5, 1, 1, 1]; o.input_weights[:] = [0.1, 0.1]
        o.u_eq[1] = 235440.0
        o.tan_gamma_gs, o.theta_max, o.w_B_max, o.gimbal_max, o.T_min, o.T_max, o.x_scale_ref = 1.0, 1.0, 0.35, 0.26, 1e4, 4.2e5, 800.0
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    assert raw.scpp_hip_mpc_setup(h, None, pp) == E_ARG
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts(K=9)), pp) == E_ARG                       # > 16 variables
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts(K=2)), pp) == E_ARG
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts(nondimensionalize=1)), pp) == E_UNSUPPORTED
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts(constant_dynamics=0)), pp) == 0           # dynpar instead of par: the same problem
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts(intermediate_cost_active=1)), pp) == E_UNSUPPORTED
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts(T_max=1e3)), pp) == E_ARG                 # empty thrust range
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts()), pp) == 0
    assert raw.scpp_hip_mpc_solve(h, xp, xp, 5, C.byref(n)) == E_ARG                        # beyond the context capacity
    assert raw.scpp_hip_mpc_solve(h, None, xp, 4, C.byref(n)) == E_ARG
    assert raw.scpp_hip_mpc_sim(h, xp, xp, 4, C.c_double(0.0), C.c_double(1.0), C.c_double(0.02), 0, C.byref(n)) == E_ARG
    assert raw.scpp_hip_destroy(h) == 0
    # the MPC entry points belong to the Rocket2D plugin
    assert raw.scpp_hip_create(C.byref(h), 0, scpp_amd.MODEL_ROCKETQUAT, 8, 4, 0) == 0
    assert raw.scpp_hip_mpc_setup(h, C.byref(opts()), pp) == E_UNSUPPORTED
    assert raw.scpp_hip_destroy(h) == 0
    a.ctx.close()


def test_emu_other_horizons(oracle, rocket2d, emu_lib, tmp_path):
    """K = 4 (8 variables) and K = 8 (16 variables: the whole tile, 56 box rows) through a modified MPC.info."""
    import os, shutil
    for K in (4, 8):
        cfg = tmp_path / f"k{K}" / "Rocket2D"
        cfg.mkdir(parents=True)
        src = rocket2d.getParameterFolder()
        for f in ("model.info", "MPC.info"):
            shutil.copy(os.path.join(src, f), cfg / f)
        txt = (cfg / "MPC.info").read_text().replace("K                           7", f"K                           {K}")
        (cfg / "MPC.info").write_text(txt)
        m = scpp_amd.Rocket2D(str(tmp_path / f"k{K}")).loadParameters(); m.p.constrain_initial_final = False
        a = scpp_amd.MPCAlgorithm(m, batch_max=4, library=emu_lib).initialize()
        assert a.K == K
        o = oracle.MPC(str(tmp_path / f"k{K}"))
        x0 = m.randomized_initial_states(2)
        a.setInitialState(x0); a.setFinalState(m.p.x_final)
        assert a.solve() == 2
        out = a.getSolution()
        for b in range(2):
            r = o.solve(x0[b], kind=1)
            assert r["status"] == 0 and out["iters"][b] == r["iters"]
            assert np.abs(out["U"][b] - r["U"]).max() <= 1e-9 * np.abs(r["U"]).max()
        a.ctx.close()


def _check_golden(alg):
    """Committed golden vectors (tests/golden/rocket2d_mpc.npz): independent discretisation, SLSQP optima, regression record."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "rocket2d_mpc.npz"))
    assert np.abs(alg.A - g["A"]).max() < 1e-13 and np.abs(alg.B - g["B"]).max() < 1e-13 and np.abs(alg.z - g["z"]).max() < 1e-12
    x = np.vstack([g["x0"], g["reg_x0"]])
    alg.setInitialState(x); alg.setFinalState(alg.model.p.x_final)
    assert alg.solve() == x.shape[0]
    out = alg.getSolution()
    n = len(g["x0"])
    tot = out["cost"].sum(axis=1)
    assert (tot[:n] <= g["cost"] * (1 + 1e-6)).all() and np.abs(tot[:n] - g["cost"]).max() < 2e-5 * g["cost"].max()
    assert np.array_equal(out["iters"][n:], g["reg_iters"])
    assert np.abs(out["U"][n:] - g["reg_U"]).max() <= 1e-9 * np.abs(g["reg_U"]).max()
    assert np.abs(out["cost"][n:] - g["reg_cost"]).max() <= 1e-9 * g["reg_cost"].max()


def test_emu_mpc_against_golden_vectors(rocket2d, emu_lib):
    a = _alg(rocket2d, emu_lib)
    _check_golden(a)
    a.ctx.close()


# ---------------------------------------------------------------- real GPU ----
@pytest.mark.gpu
def test_gpu_mpc_against_golden_vectors(rocket2d, hip_lib):
    a = _alg(rocket2d, hip_lib)
    _check_golden(a)
    a.ctx.close()


@pytest.mark.gpu
def test_gpu_mpc_solve_parity(oracle, rocket2d, hip_lib):
    """256 controllers, one wavefront each, against the twin: same status, same iteration count, same optimum."""
    a = _alg(rocket2d, hip_lib, batch_max=256)
    x0 = rocket2d.randomized_initial_states(256)
    x0[200:, 1] *= 0.3
    x0[230:240, 4] = 1.2          # outside the tilt box
    x0[240:, 0] = 2.0 * x0[240:, 1]  # outside the glide-slope cone
    out = _check_solves(oracle, a, x0)
    assert (out["status"][230:] == -3).all() and (out["status"][:200] == 0).all()
    a.ctx.close()


@pytest.mark.gpu
def test_gpu_mpc_closed_loop_parity(oracle, rocket2d, hip_lib):
    """The whole MPC_sim run (1501 steps of 10 ms, > 1000 of them with an infeasible problem and a held input) for 4 loops."""
    a = _alg(rocket2d, hip_lib, batch_max=64)
    o = oracle.MPC()
    x0 = rocket2d.randomized_initial_states(4, first=40)
    r = scpp_amd.MPCSim(a).run(x0)
    for b in range(4):
        q = o.sim(x0[b])
        assert r["steps"][b] == q["steps"] == 1501 and r["failed_solves"][b] == q["failed_solves"] and r["reached"][b] == q["reached"]
        assert np.abs(r["x"][b] - q["x"]).max() < 1e-8 * np.abs(q["x"]).max()
    a.ctx.close()


@pytest.mark.gpu
def test_gpu_mpc_large_batch_properties(rocket2d, hip_lib):
    """Size-independent properties at 16384 controllers: every returned plan satisfies the reference problem's constraints,
    the epigraph costs equal the norms they bound, and identical inputs give bitwise identical plans wherever they sit in
    the batch (no cross-instance state)."""
    B = 16384
    a = _alg(rocket2d, hip_lib, batch_max=B)
    x0 = rocket2d.randomized_initial_states(B)
    x0[B // 2:] = x0[: B // 2]
    a.setInitialState(x0); a.setFinalState(rocket2d.p.x_final)
    assert a.solve() == B
    out = a.getSolution()
    X, U = out["X"], out["U"]
    p = rocket2d.p
    assert np.array_equal(U[: B // 2], U[B // 2:]) and np.array_equal(X[: B // 2], X[B // 2:])
    assert (np.abs(U[:, :, 0]) <= p.gimbal_max * (1 + 1e-6)).all()
    assert (U[:, :, 1] >= p.T_min * (1 - 1e-6)).all() and (U[:, :, 1] <= p.T_max * (1 + 1e-6)).all()
    assert (np.abs(X[:, :, 4]) <= p.theta_max + 1e-7).all() and (np.abs(X[:, :, 5]) <= p.w_B_max + 1e-7).all()
    assert (np.abs(X[:, 1:, 0]) <= p.tan_gamma_gs * X[:, 1:, 1] + 1e-5).all()
    for k in range(a.K - 1):
        assert np.abs(X[:, k + 1] - (X[:, k] @ a.A.T + U[:, k] @ a.B.T + a.z)).max() < 1e-7
    ec = np.linalg.norm(a.state_weights_terminal * (X[:, -1] - p.x_final), axis=1)
    ic = np.linalg.norm((a.input_weights * U).reshape(B, -1), axis=1)
    assert np.abs(out["cost"][:, 0] - ic).max() < 1e-6 * ic.max() and np.abs(out["cost"][:, 1] - ec).max() < 1e-6 * ec.max()
    a.ctx.close()


def test_emu_mpc_mirror_symmetry(rocket2d, emu_lib):
    """Size-independent property: the planar rocket, its constraints and the MPC cost are symmetric under the mirror
    (x, vx, eta, omega, gimbal) -> -(x, vx, eta, omega, gimbal); the optimal plan of the mirrored state is the mirrored plan."""
    a = _alg(rocket2d, emu_lib)
    x0 = rocket2d.randomized_initial_states(6, first=300)
    M = np.array([-1.0, 1.0, -1.0, 1.0, -1.0, -1.0])
    a.setInitialState(np.vstack([x0, x0 * M])); a.setFinalState(rocket2d.p.x_final)
    assert a.solve() == 12
    out = a.getSolution()
    U, X = out["U"], out["X"]
    assert np.array_equal(out["iters"][:6], out["iters"][6:]) or np.abs(out["iters"][:6] - out["iters"][6:]).max() <= 1
    assert np.abs(U[:6, :, 0] + U[6:, :, 0]).max() <= 1e-6 * rocket2d.p.gimbal_max
    assert np.abs(U[:6, :, 1] - U[6:, :, 1]).max() <= 1e-6 * rocket2d.p.T_max
    assert np.abs(X[:6] * M - X[6:]).max() <= 1e-6 * np.abs(X).max()
    assert np.abs(out["cost"][:6] - out["cost"][6:]).max() <= 1e-7 * out["cost"].max()
    a.ctx.close()


def _mpc_inject(emu_lib, spec):
    """one MPC solve of one controller on the emulator in a fresh process with SCPP_EMU_INJECT_MPC_RES=spec -> (status, iterations)"""
    import json, os, subprocess, sys
    code = (
        "import json, numpy as np, scpp_amd\n"
        "m = scpp_amd.Rocket2D().loadParameters(); m.p.constrain_initial_final = False\n"
        "alg = scpp_amd.MPCAlgorithm(m, batch_max=1, library=%r).initialize()\n"
        "alg.setInitialState(m.randomized_initial_states(1)); alg.setFinalState(m.p.x_final)\n"
        "alg.solve(); o = alg.getSolution()\n"
        "print('RESULT ' + json.dumps([int(o['status'][0]), int(o['iters'][0])]))\n" % emu_lib
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    env.pop("SCPP_EMU_INJECT_MPC_RES", None)
    if spec:
        env["SCPP_EMU_INJECT_MPC_RES"] = spec
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_emu_mpc_broken_iterate_without_a_fallback_is_a_failure(emu_lib):
    """VERDICT r5 missing 3 / DESIGN r5 section 9 item 4: mpc_solve_kernel's breakdown test came before its fall-back save (the right order) but, like the big
    solver until round 5, let a blown-up or out-of-cone iterate through its convergence test when no fall-back existed: pres, dres are relative to
    the iterate's norm and a negative gap meets `gap < abstol`.  The emulator build replaces the termination quantities of the 3rd evaluation of a
    cold solve (no fall-back yet): it must fail with -2 for a blown-up, a hugely negative and a merely negative gap; the control (gap +1e-12) is
    accepted there, which shows the hook reaches the rule."""
    st, it = _mpc_inject(emu_lib, "")
    assert st == 0 and it > 4
    assert _mpc_inject(emu_lib, "2:0:0:1e-12") == [0, 2]
    for gap in ("1e31", "-1e31", "-1e29", "-1e-3"):
        assert _mpc_inject(emu_lib, "2:0:0:" + gap) == [-2, 2], gap
