"""The split schedule of the interior-point solve (csrc/ipm_split.h: two kernels per interior-point iteration, the factor sweep at two
wavefronts per SIMD and everything else at three, state carried through the workspace between launches) against the resident kernel
(one launch per solve) on the CPU wave emulation of the same sources: the two drive the SAME phase functions in the same order, so
every output must be BITWISE identical -- trajectories, counters, statuses, the interior-point iteration counts -- in SCvx mode (batch and
streaming entry points, rejections and warm starts included), in SC mode (sigma border: two-column factor sweeps), and with too few launch
pairs the instance must be reported, not silently returned half-solved."""
import numpy as np
import pytest

import scpp_amd
from scpp_amd import _lib

KEYS = ("X", "U", "sigma", "nu_norm", "sc_iters", "solves", "converged", "status", "ipm_iters")


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a


def _scvx(model, emu_lib, schedule, K, B, maxit, pairs=0, stream=False):
    alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=B, library=emu_lib, max_iterations=maxit).initialize()
    alg.ctx.set_ipm_schedule(schedule, pairs)
    x0 = model.randomized_initial_states(B, first=100)
    if stream:
        alg.solveStream(np.concatenate([x0, model.randomized_initial_states(B, first=300)]), slots=B, pools=2)
        return alg.ctx.stream_download_rows()
    alg.solve(x0)
    return alg.getSolution()


@pytest.mark.parametrize("K,B,maxit", [(8, 4, 5), (15, 3, 4)])
def test_split_schedule_is_bitwise_the_resident_kernel_scvx(model, emu_lib, K, B, maxit):
    ref = _scvx(model, emu_lib, _lib.IPM_RESIDENT, K, B, maxit)
    ws = _scvx(model, emu_lib, _lib.IPM_RESIDENT_WS, K, B, maxit)
    sp = _scvx(model, emu_lib, _lib.IPM_SPLIT, K, B, maxit)
    assert ref["solves"].sum() > B * 2 and ref["ipm_iters"].min() > 10  # real work, incl. warm-started re-solves
    for key in KEYS:
        assert np.array_equal(_bits(ref[key]), _bits(ws[key])), ("workspace-resident segment fields", key)
        assert np.array_equal(_bits(ref[key]), _bits(sp[key])), ("split schedule", key)


def test_split_schedule_streaming_engine(model, emu_lib):
    ref = _scvx(model, emu_lib, _lib.IPM_RESIDENT, 8, 4, 4, stream=True)
    sp = _scvx(model, emu_lib, _lib.IPM_SPLIT, 8, 4, 4, stream=True)
    assert ref.shape == sp.shape and np.array_equal(_bits(ref), _bits(sp))


def test_split_schedule_sc_mode_two_column_factor_sweeps(model, emu_lib):
    """SCAlgorithm mode (free final time): the factor sweep carries the sigma border column, the spec travels through the resume block"""
    outs = []
    for schedule in (_lib.IPM_RESIDENT, _lib.IPM_SPLIT):
        alg = scpp_amd.SCAlgorithm(model, K=8, batch_max=3, library=emu_lib).initialize()
        alg.ctx.set_ipm_schedule(schedule)
        alg.solve(model.randomized_initial_states(3, first=7))
        outs.append(alg.getSolution())
    for key in ("X", "U", "sigma", "nu_norm", "sc_iters", "converged", "status", "ipm_iters"):
        assert np.array_equal(_bits(outs[0][key]), _bits(outs[1][key])), key
    assert outs[0]["ipm_iters"].min() > 20


def test_split_schedule_argument_checks(model, emu_lib):
    ctx = scpp_amd.Context(K=8, batch_max=2, library=emu_lib)
    for bad in ((-1, 0), (3, 0), (1, -1)):
        with pytest.raises(scpp_amd.ScppHipError):
            ctx.set_ipm_schedule(*bad)
    ctx.set_ipm_schedule(_lib.IPM_SPLIT, 7)
    ctx.set_ipm_schedule(_lib.IPM_RESIDENT)


@pytest.mark.parametrize("K,N,S,maxit", [(8, 7, 3, 4), (15, 5, 2, 3)])
def test_persistent_stream_engine_rows_are_bitwise_the_pool_engine(model, emu_lib, K, N, S, maxit):
    """scpp_hip_set_stream_engine(SCPP_STREAM_PERSISTENT): ONE launch, a wavefront per slot walks its instances through refill ->
    multipleShooting of every segment -> sub-problem solve -> cost / accept / reject (csrc/scvx_persistent.h).  Same bodies, same order, same
    buffers as the pool engine's four kernels per round: every result row must be bitwise the pool engine's, and the batch entry point's."""
    x0 = model.randomized_initial_states(N, first=40)
    rows = []
    for engine in (_lib.STREAM_POOLS, _lib.STREAM_PERSISTENT):
        alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=S, library=emu_lib, max_iterations=maxit).initialize()
        alg.ctx.set_stream_engine(engine)
        n = alg.solveStream(x0, slots=S, pools=2)
        r = alg.ctx.stream_download_rows()
        assert n == int(scpp_amd.Context.unpack_stream_rows(r, K)["converged"].sum())
        rows.append(r)
    assert rows[0].shape == rows[1].shape == (N, K * 18 + 10) and np.array_equal(_bits(rows[0]), _bits(rows[1]))
    got = scpp_amd.Context.unpack_stream_rows(rows[1], K)
    assert (got["instance"] == np.arange(N)).all() and got["solves"].sum() > 2 * N
    alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=N, library=emu_lib, max_iterations=maxit).initialize()
    alg.solve(x0)
    ref = alg.getSolution()
    for key in KEYS:
        assert np.array_equal(_bits(got[key]), _bits(ref[key])), key


def test_persistent_sc_loop_is_bitwise_the_launch_loop(model, emu_lib, monkeypatch):
    """scpp_hip_sc_solve on the persistent kernel (one launch: every wavefront takes its instance through up to max_iterations rounds of
    multipleShooting + solve, csrc/scvx_persistent.h: sc_persistent_kernel) against the loop of launches (scpp_hip_set_stream_engine(POOLS) selects
    it): same bodies, same order -- bitwise.  The library gives only COLD solves of >= 3072 instances to the persistent kernel (that is where it is
    faster, profiles/r05_ab_persistent_batch_sizes.json); SCPP_SC_PERSISTENT_MIN lowers the bar for this test, and the warm-started second solve runs
    the loop of launches on both sides, on the state the first solve left."""
    monkeypatch.setenv("SCPP_SC_PERSISTENT_MIN", "1")
    outs = []
    for engine in (_lib.STREAM_POOLS, _lib.STREAM_PERSISTENT):
        alg = scpp_amd.SCAlgorithm(model, K=8, batch_max=4, library=emu_lib).initialize()
        alg.ctx.set_stream_engine(engine)
        x0 = model.randomized_initial_states(4, first=21)
        alg.solve(x0)
        a = alg.getSolution()
        alg.solve(x0 * (1.0 + 1e-3 * np.arange(4)[:, None] * (np.arange(14) == 3)), warm_start=True)
        b = alg.getSolution()
        outs.append((a, b))
    for i in range(2):
        for key in ("X", "U", "sigma", "nu_norm", "sc_iters", "converged", "status", "ipm_iters"):
            assert np.array_equal(_bits(outs[0][i][key]), _bits(outs[1][i][key])), (i, key)
    assert outs[0][0]["ipm_iters"].min() > 20


def test_persistent_cost_step_with_two_passes_of_segment_pairs(model, emu_lib):
    """K = 34: 33 segments = one full pass of 32 lane pairs + a second pass with a single pair (scvxCostUpdateSplit takes the K - 1 segments through in
    passes of 32 pairs; the BASELINE's K = 50 is two passes as well, but only the GPU suite runs it).  Persistent kernel against the pool engine, rows
    bitwise."""
    K, N, S, maxit = 34, 3, 2, 3
    x0 = model.randomized_initial_states(N, first=60)
    rows = []
    for engine in (_lib.STREAM_POOLS, _lib.STREAM_PERSISTENT):
        alg = scpp_amd.SCvxAlgorithm(model, K=K, batch_max=S, library=emu_lib, max_iterations=maxit).initialize()
        alg.ctx.set_stream_engine(engine)
        alg.solveStream(x0, slots=S)
        rows.append(alg.ctx.stream_download_rows())
        assert alg.ctx.stream_rounds()["pools"] == (0 if engine == _lib.STREAM_PERSISTENT else 1)
    assert np.array_equal(_bits(rows[0]), _bits(rows[1]))
    got = scpp_amd.Context.unpack_stream_rows(rows[1], K)
    assert (got["status"] == 0).all() and got["solves"].sum() >= N * maxit and (got["nonlinear_cost"] > 0).all()


def test_truncated_split_schedule_reports_unfinished_instances_as_failures(model, emu_lib):
    """with fewer (factor, rest) launch pairs than 2 maxit + 1 an instance can still be waiting for a factor sweep when the last launch
    ends; it has written no outputs.  The finalising pass (ipm_split.h: ipm_split_finalize_kernel) retires it as an iteration-limit failure: status -1,
    SC loop stopped, iteration counted -- not the previous solve's rows with its status 0."""
    alg = scpp_amd.SCAlgorithm(model, K=8, batch_max=2, library=emu_lib).initialize()
    x0 = model.randomized_initial_states(2, first=3)
    alg.ctx.sc_setup(model.p, alg.opts, x0)
    alg.ctx.sc_iterate()  # a complete solve first (resident schedule): status 0 is what a stale row would show
    first = alg.ctx.download()
    assert (first["status"] == 0).all() and (first["sc_iters"] == 1).all()
    alg.ctx.set_ipm_schedule(_lib.IPM_SPLIT, 3)  # 3 factor sweeps: far too few for a solve of 15+ iterations
    alg.ctx.sc_iterate()
    out = alg.ctx.download()
    assert (out["status"] == -1).all(), out["status"]
    assert (out["sc_iters"] == 2).all()
    assert np.array_equal(_bits(out["X"]), _bits(first["X"]))  # the iterate of the last COMPLETE solve stays
    # the full schedule on a fresh context is unaffected (no finalising launch) and equals the resident kernel (tests above)
