"""Host logic: ParameterServer semantics (parameterServer.hpp:64-127), model loading, counter RNG."""
import os

import numpy as np
import pytest

import scpp_amd


def test_parameter_server_semantics(tmp_path):
    f = tmp_path / "a.info"
    f.write_text("; comment\nfoo 1.5 ; trailing\nflag true\nvec\n{\n  scaling 2.\n  (0) 1.\n  (1) 2. ; c\n}\n")
    ps = scpp_amd.ParameterServer(str(f))
    assert ps.load_scalar("foo") == 1.5
    assert ps.load_scalar("flag", bool) is True
    assert ps.load_vector("vec", 2) == [2.0, 4.0]
    with pytest.raises(RuntimeError, match="Missing entries"):
        ps.load_vector("vec", 3)
    with pytest.raises(RuntimeError, match="Redundant entries"):
        ps.load_vector("vec", 1)
    with pytest.raises(RuntimeError, match="Failed to load scalar"):
        ps.load_scalar("nope")


def test_rocketquat_loader_matches_oracle(oracle, model):
    sc = oracle.SC(oracle.ROCKETQUAT, K=50)
    assert np.allclose(model.x_init, sc.x_init(), rtol=0, atol=0)
    assert np.allclose(list(model.p.x_final), sc.x_final(), rtol=0, atol=0)
    assert abs(model.p.alpha_m - 1.0 / (275.0 * 9.81)) < 1e-18
    opts = scpp_amd.load_sc_opts(model.getParameterFolder())
    assert (opts.K, opts.max_iterations, opts.weight_trust_region_trajectory, opts.weight_virtual_control) == (50, 15, 50.0, 1000.0)


def test_randomised_initial_states_match_oracle(oracle, model):
    X = model.randomized_initial_states(5, seed=20260927, first=3)
    for b in range(5):
        sc = oracle.SC(oracle.ROCKETQUAT, K=50)
        sc.randomize(20260927, 3 + b)
        assert np.array_equal(X[b], sc.x_init())
    # r_z, mass and body rates untouched; quaternion normalised
    assert np.all(X[:, 0] == 24000.0) and np.all(X[:, 3] == 800.0) and np.all(X[:, 11:] == 0.0)
    assert np.allclose(np.linalg.norm(X[:, 7:11], axis=1), 1.0)


def test_missing_hip_library_fails_loudly(tmp_path):
    with pytest.raises(scpp_amd.ScppHipError, match="no CPU fallback"):
        scpp_amd.load_library(str(tmp_path / "libscpp_hip.so"))
