"""Oracle flow maps / Jacobians against the sympy golden vectors (G1) and finite differences."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.mark.parametrize("name,model_id", [("rocketquat", 0), ("rocket2d", 1)])
def test_flow_and_jacobians_match_sympy_golden(oracle, name, model_id):
    g = np.load(os.path.join(GOLDEN, f"{name}_jacobians.npz"))
    for i in range(g["x"].shape[0]):
        f, A, B = oracle.flow(model_id, g["x"][i], g["u"][i], g["par"][i])
        sc = max(1.0, np.abs(g["A"][i]).max(), np.abs(g["B"][i]).max())
        assert np.abs(f - g["f"][i]).max() <= 1e-13 * max(1.0, np.abs(g["f"][i]).max())
        assert np.abs(A - g["A"][i]).max() <= 1e-13 * sc
        assert np.abs(B - g["B"][i]).max() <= 1e-13 * sc


def test_rocketquat_quirks(oracle):
    """gyroscopic term w x w == 0 (no dependence of w-dot on w); un-normalised rotation matrix."""
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, 14); x[0] = 1.3
    u = rng.uniform(0.1, 1, 4)
    par = rng.uniform(0.5, 2, 10)
    f, A, B = oracle.flow(0, x, u, par)
    assert np.all(A[11:14, 11:14] == 0.0)
    x2 = x.copy(); x2[7:11] *= 2.0  # scaling q changes R(q) because it is NOT normalised
    f2, _, _ = oracle.flow(0, x2, u, par)
    assert np.abs(f2[4:7] - f[4:7]).max() > 1e-3


def test_jacobian_finite_difference(oracle):
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, 14); x[0] = 1.1
    u = rng.uniform(0.2, 1, 4)
    par = rng.uniform(0.5, 2, 10)
    f, A, B = oracle.flow(0, x, u, par)
    h = 1e-6
    for j in range(14):
        xp, xm = x.copy(), x.copy(); xp[j] += h; xm[j] -= h
        fd = (oracle.flow(0, xp, u, par)[0] - oracle.flow(0, xm, u, par)[0]) / (2 * h)
        assert np.abs(fd - A[:, j]).max() < 1e-7
    for j in range(4):
        up, um = u.copy(), u.copy(); up[j] += h; um[j] -= h
        fd = (oracle.flow(0, x, up, par)[0] - oracle.flow(0, x, um, par)[0]) / (2 * h)
        assert np.abs(fd - B[:, j]).max() < 1e-7
