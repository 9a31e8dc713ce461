"""Error behaviour and edge sizes of the C ABI (include/scpp_hip.h): every call returns SCPP_OK or a negative code,
never throws / exits.  Runs on the CPU wave-emulation build of the same sources (host logic is identical)."""
import ctypes as C

import numpy as np
import pytest

import scpp_amd
from scpp_amd import _lib

E_ARG, E_HIP, E_UNSUPPORTED, E_STATE = -1, -2, -3, -4


@pytest.fixture(scope="module")
def raw(emu_lib):
    return _lib.load_library(emu_lib)


def _create(raw, K=8, B=4, model=0):
    h = C.c_void_p()
    rc = raw.scpp_hip_create(C.byref(h), 0, model, K, B, 0)
    return rc, h


def test_create_argument_checks(raw):
    h = C.c_void_p()
    assert raw.scpp_hip_create(None, 0, 0, 8, 4, 0) == E_ARG
    assert raw.scpp_hip_create(C.byref(h), 0, 7, 8, 4, 0) == E_ARG          # unknown model
    assert raw.scpp_hip_create(C.byref(h), 0, 0, 2, 4, 0) == E_ARG          # K < 3
    assert raw.scpp_hip_create(C.byref(h), 0, 0, 8, 0, 0) == E_ARG          # empty batch capacity
    assert raw.scpp_hip_create(C.byref(h), 0, 0, 65, 4, 0) in (E_ARG, E_UNSUPPORTED)  # one wavefront lane per stage: K <= 64
    rc, h = _create(raw)
    assert rc == 0 and h.value
    assert raw.scpp_hip_destroy(h) == 0
    assert raw.scpp_hip_destroy(None) in (0, E_ARG)


def test_state_and_argument_errors(raw, model):
    rc, h = _create(raw, K=8, B=4)
    assert rc == 0
    n = C.c_int(0)
    # SC calls before sc_setup
    assert raw.scpp_hip_sc_iterate(h, C.byref(n)) == E_STATE
    assert raw.scpp_hip_sc_solve(h, C.byref(n)) == E_STATE
    assert raw.scpp_hip_sc_finish(h, C.byref(n)) == E_STATE
    assert raw.scpp_hip_socp_solve(h) == E_STATE
    mask = np.ones(4, dtype=np.int32)
    assert raw.scpp_hip_sc_set_active(h, mask.ctypes.data_as(C.c_void_p), 4) in (E_STATE, E_ARG)
    opts = scpp_amd.load_sc_opts(model.getParameterFolder(), 8)
    x0 = np.ascontiguousarray(model.randomized_initial_states(4))
    xp = x0.ctypes.data_as(C.c_void_p)
    # batch larger than the context capacity, empty batch, null pointers
    assert raw.scpp_hip_sc_setup(h, C.byref(model.p), C.byref(opts), xp, 5, 0) == E_ARG
    assert raw.scpp_hip_sc_setup(h, C.byref(model.p), C.byref(opts), xp, 0, 0) == E_ARG
    assert raw.scpp_hip_sc_setup(h, None, C.byref(opts), xp, 4, 0) == E_ARG
    # warm start without a previous solve
    assert raw.scpp_hip_sc_setup(h, C.byref(model.p), C.byref(opts), xp, 4, 1) == E_STATE
    # configurations the device solver does not implement are refused, not silently mis-solved
    # (free_final_time false and zero-order hold are implemented since round 3; roll control is not: 18 stage variables)
    rollp = type(model.p).from_buffer_copy(model.p)
    rollp.enable_roll_control = 1
    assert raw.scpp_hip_sc_setup(h, C.byref(rollp), C.byref(opts), xp, 4, 0) == E_UNSUPPORTED
    bad = scpp_amd.load_sc_opts(model.getParameterFolder(), 9)  # K differs from the context's
    assert raw.scpp_hip_sc_setup(h, C.byref(model.p), C.byref(bad), xp, 4, 0) == E_UNSUPPORTED
    # a good setup, then a mask of the wrong length
    assert raw.scpp_hip_sc_setup(h, C.byref(model.p), C.byref(opts), xp, 4, 0) == 0
    assert raw.scpp_hip_sc_set_active(h, mask.ctypes.data_as(C.c_void_p), 3) == E_ARG
    assert raw.scpp_hip_sc_set_active(h, mask.ctypes.data_as(C.c_void_p), 4) == 0
    # warm start with a different batch size
    assert raw.scpp_hip_sc_setup(h, C.byref(model.p), C.byref(opts), xp, 3, 1) == E_STATE
    assert raw.scpp_hip_destroy(h) == 0


def test_all_instances_masked_out_is_a_no_op(model, emu_lib):
    alg = scpp_amd.SCAlgorithm(model, K=8, batch_max=3, library=emu_lib).initialize()
    x0 = model.randomized_initial_states(3)
    alg.ctx.sc_setup(model.p, alg.opts, x0)
    alg.ctx.sc_set_active(np.zeros(3, dtype=np.int32))
    assert alg.ctx.sc_solve() == 0
    out = alg.ctx.download()
    assert (out["sc_iters"] == 0).all() and (out["status"] == 0).all()


@pytest.mark.parametrize("K,B", [(3, 1), (4, 2), (5, 3)])
def test_smallest_horizons_and_ragged_batches(oracle, model, emu_lib, K, B):
    """K = 3 is the smallest horizon the library accepts; odd batch sizes exercise the XCD-aware
    block map of the discretisation (8 instance slots per group) with partially filled groups."""
    alg = scpp_amd.SCAlgorithm(model, K=K, batch_max=B, library=emu_lib).initialize()
    x0 = model.randomized_initial_states(B)
    alg.ctx.sc_setup(model.p, alg.opts, x0)
    alg.ctx.sc_iterate()
    out = alg.ctx.download()
    for b in range(B):
        sc = oracle.SC(oracle.ROCKETQUAT, K=K); sc.randomize(20260927, b); sc.set_solver(1); sc.solve()
        X1, U1, t1 = sc.iterate(1)
        if sc.info()[0][5] != 0:  # oracle's own solver status for the first sub-problem
            continue
        assert out["status"][b] == 0
        assert np.abs(out["X"][b] - X1).max() <= 1e-7 * max(1.0, np.abs(X1).max())
        assert abs(out["sigma"][b] - t1) <= 1e-7 * abs(t1)


def test_empty_jobs_are_refused_not_ignored(model, emu_lib):
    """a job of zero instances is an argument error at every batched entry point (SCPP_E_ARG), never a silent no-op"""
    alg = scpp_amd.SCvxAlgorithm(model, K=8, batch_max=2, library=emu_lib, max_iterations=2).initialize()
    none = model.randomized_initial_states(2)[:0]
    for call in (lambda: alg.solveStream(none, slots=2, pools=0), lambda: alg.solve(none)):
        with pytest.raises(scpp_amd.ScppHipError, match="code -1"):
            call()
    sc = scpp_amd.SCAlgorithm(model, K=8, batch_max=2, library=emu_lib).initialize()
    with pytest.raises(scpp_amd.ScppHipError, match="code -1"):
        sc.ctx.sc_setup(model.p, sc.opts, none)


def test_roll_control_is_refused_by_every_entry_point(raw, model):
    """rocketQuat.cpp:135-138 (`enable_roll_control`): a reference code path this engine permanently does not run (include/scpp_hip.h:
    18 free variables per node against one 16-wide tile).  All THREE entry points that take the model parameters say so with
    SCPP_E_UNSUPPORTED instead of solving a different problem: scpp_hip_sc_setup, scpp_hip_scvx_setup, scpp_hip_scvx_solve_stream."""
    rc, h = _create(raw, K=8, B=4)
    assert rc == 0
    rollp = type(model.p).from_buffer_copy(model.p)
    rollp.enable_roll_control = 1
    x0 = np.ascontiguousarray(model.randomized_initial_states(4))
    xp = x0.ctypes.data_as(C.c_void_p)
    sc = scpp_amd.load_sc_opts(model.getParameterFolder(), 8)
    vx = scpp_amd.load_scvx_opts(model.getParameterFolder(), 8)
    n = C.c_int(-7)
    assert raw.scpp_hip_sc_setup(h, C.byref(rollp), C.byref(sc), xp, 4, 0) == E_UNSUPPORTED
    assert raw.scpp_hip_scvx_setup(h, C.byref(rollp), C.byref(vx), xp, 4, 0) == E_UNSUPPORTED
    assert raw.scpp_hip_scvx_solve_stream(h, C.byref(rollp), C.byref(vx), xp, 4, 4, 1, C.byref(n)) == E_UNSUPPORTED
    # nothing was set up by the refused calls
    assert raw.scpp_hip_scvx_solve(h, C.byref(n)) == E_STATE and raw.scpp_hip_sc_solve(h, C.byref(n)) == E_STATE
    # ... and the same parameters with roll control off are accepted
    assert raw.scpp_hip_scvx_setup(h, C.byref(model.p), C.byref(vx), xp, 4, 0) == 0
    assert raw.scpp_hip_destroy(h) == 0


def test_query_reports_the_build_constants(raw):
    """scpp_hip_query: a binding asks the library for its build-defined constants instead of hard-coding them (the rejection-cap
    status moved from -4 to -5 with no way to notice); load_library refuses a library whose answers differ from the binding's."""
    v = C.c_longlong(0)
    for what, want in ((_lib.Q_ABI_REVISION, _lib.ABI_REVISION), (_lib.Q_STATUS_REJECTION_CAP, scpp_amd.STATUS_REJECTION_CAP),
                       (_lib.Q_SCVX_SOLVE_CAP, scpp_amd.SCVX_SOLVE_CAP), (_lib.Q_MAX_K, 64), (_lib.Q_MPC_MAX_K, 8)):
        assert raw.scpp_hip_query(what, C.byref(v)) == 0 and v.value == want
    assert raw.scpp_hip_query(99, C.byref(v)) == E_ARG and raw.scpp_hip_query(0, None) == E_ARG
    assert scpp_amd.STATUS_REJECTION_CAP not in (0, E_ARG, E_HIP, E_UNSUPPORTED, E_STATE)  # a status, distinct from every return code
    h = C.c_void_p()
    assert raw.scpp_hip_create(C.byref(h), 0, 0, _lib.query(_lib.Q_MAX_K, raw) + 1, 4, 0) == E_ARG
