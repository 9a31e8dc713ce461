"""Seeded differential test of the streaming engine (scpp_hip_scvx_solve_stream: slot pools, refills, roll-backs, per-pool views of
every buffer) against the plain batch entry point on the CPU wave emulator: random (model, hold, K, instances, slots, pools,
iteration cap) -- the defect of round 3 (one pool view offset by the wrong model's sizes) lived exactly in a
combination no hand-written case had.  Every row of the stream must be BITWISE the row of the batch run: instances are independent and
both entry points run the same kernels (SCvxAlgorithm.cpp:61-164 per instance, whatever slot it occupies).
Also covered: slots > instances, one slot, more pools than slots can fill, instance counts that are not multiples of anything.
Power of the test: with the round-3 defect put back (`v.Xold += f * K * 14; v.Uold += f * K * 4` in scvxBuffersRange) three of the twelve
cases fail (the Rocket2D ones with more than one pool and a rejected candidate)."""
import os
import shutil

import numpy as np
import pytest

import scpp_amd

KEYS = ("X", "U", "sigma", "nu_norm", "nonlinear_cost", "trust_region", "sc_iters", "solves", "converged", "status", "ipm_iters")


def _config(tmp_path, hold_foh):
    cfg = tmp_path / ("cfg_foh" if hold_foh else "cfg_zoh")
    if not cfg.exists():
        shutil.copytree(os.path.join(os.path.dirname(scpp_amd.__file__), "config"), cfg)
        for mdl in ("RocketQuat", "Rocket2D"):
            p = cfg / mdl / "SCvx.info"
            t = p.read_text()
            if not hold_foh:
                assert "interpolate_input                   true" in t
                t = t.replace("interpolate_input                   true", "interpolate_input                   false")
            if mdl == "Rocket2D":  # the nondimensionalised Rocket2D configuration converges and rejects candidates
                t = t.replace("nondimensionalize                   false", "nondimensionalize                   true")
            p.write_text(t)
    return str(cfg)


def _cases(n, seed, big=False):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        model = ("RocketQuat", "Rocket2D")[int(rng.integers(2))]
        foh = bool(rng.integers(2))
        K = int(rng.integers(10, 51)) if big else int(rng.integers(5, 10))
        N = int(rng.integers(20, 200)) if big else int(rng.integers(2, 10))
        slots = int(rng.integers(8, N + 10)) if big else int(rng.integers(1, N + 3))
        pools = int(rng.integers(1, 5))
        maxit = int(rng.integers(3, 9)) if big else int(rng.integers(3, 7))
        out.append((i, model, foh, K, N, slots, pools, maxit))
    if big:
        return out
    # hand-picked corners: one slot; as many pools as slots; slots beyond the instance count; a single instance through several pools
    out += [(n, "Rocket2D", True, 6, 5, 1, 1, 4), (n + 1, "RocketQuat", False, 5, 7, 3, 3, 3), (n + 2, "Rocket2D", False, 7, 3, 9, 4, 5),
            (n + 3, "RocketQuat", True, 6, 1, 4, 3, 4)]
    return out


_ID = lambda c: "%d-%s-%s-K%d-N%d-s%d-p%d-it%d" % (c[0], c[1], "foh" if c[2] else "zoh", *c[3:])  # noqa: E731


@pytest.mark.parametrize("case", _cases(8, 20260928), ids=_ID)
def test_emu_stream_equals_batch_on_random_configurations(emu_lib, tmp_path, case):
    _stream_case(emu_lib, tmp_path, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(10, 20260929, big=True), ids=_ID)
def test_stream_equals_batch_on_random_configurations_on_gpu(hip_lib, tmp_path, case):
    """the same property on hardware at K up to 50, 20 .. 200 instances through 8 .. N + 10 slots in 1 .. 4 pools"""
    _stream_case(hip_lib, tmp_path, case)


def _stream_case(emu_lib, tmp_path, case):
    i, model, foh, K, N, slots, pools, maxit = case
    cfg = _config(tmp_path, foh)
    m = (scpp_amd.RocketQuat if model == "RocketQuat" else scpp_amd.Rocket2D)(cfg).loadParameters()
    x0 = m.randomized_initial_states(N, seed=977 + i, first=31 * i)
    ref = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=N, library=emu_lib, max_iterations=maxit).initialize()
    assert ref.opts.interpolate_input == int(foh)
    nref = ref.solve(x0)
    r = ref.getSolution()
    ref.ctx.close()
    alg = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=slots, library=emu_lib, max_iterations=maxit).initialize()
    n = alg.solveStream(x0, slots=slots, pools=pools)
    o = alg.getStreamSolution()
    rounds = alg.ctx.stream_rounds()
    assert 1 <= rounds["pools"] <= pools
    assert n == nref and (o["instance"] == np.arange(N)).all()
    for key in KEYS:
        assert np.array_equal(o[key], r[key]), (case, key)
    # pools = 0: the default engine -- the persistent kernel (csrc/scvx_persistent.h: one launch, a wavefront per slot), instantiated for both
    # models and both input holds since round 6 (the batch run above went through it as well: the rows of the POOL job were just compared with
    # it) -- on the same context, after the pool job: the same rows again
    n0 = alg.solveStream(x0, slots=slots, pools=0)
    o0 = alg.getStreamSolution()
    assert alg.ctx.stream_rounds()["pools"] == 0
    assert n0 == nref and (o0["instance"] == np.arange(N)).all()
    for key in KEYS:
        assert np.array_equal(o0[key], r[key]), (case, key, "default engine")
    # a second job on the same context (buffers re-used, queue counters reset) gives the same rows again
    if i % 3 == 0:
        n2 = alg.solveStream(x0[::-1].copy(), slots=slots, pools=pools)
        o2 = alg.getStreamSolution()
        assert n2 == nref
        for key in KEYS:
            assert np.array_equal(o2[key], r[key][::-1]), (case, key, "second job")
    alg.ctx.close()
