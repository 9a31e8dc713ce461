"""Oracle multipleShooting (reference formulation: Phi^-1, RKF78 x 5) against the DOP853 goldens (G2) and the
linearisation identity (G3)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.mark.parametrize("K", [15, 50])
def test_dd_matches_dop853_golden(oracle, K):
    g = np.load(os.path.join(GOLDEN, f"rocketquat_dd_K{K}.npz"))
    A, B, C, S, Z = oracle.discretize(0, g["par"], g["X"], g["U"], float(g["t"]))
    for name, a, o in (("A", A, g["A"]), ("B", B, g["B"]), ("C", C, g["C"]), ("S", S, g["S"]), ("Z", Z, g["Z"])):
        assert np.abs(a - o).max() <= 1e-10 * max(1.0, np.abs(o).max()), name


def test_linearisation_identity(oracle):
    """x_prop(dt) == A x_k + B u_k + C u_{k+1} + s sigma + z at the linearisation point (G3)."""
    g = np.load(os.path.join(GOLDEN, "rocketquat_dd_K50.npz"))
    X, U, t = g["X"], g["U"], float(g["t"])
    A, B, C, S, Z = oracle.discretize(0, g["par"], X, U, t)
    K = X.shape[0]
    for k in range(K - 1):
        lin = A[k] @ X[k] + B[k] @ U[k] + C[k] @ U[k + 1] + S[k] * t + Z[k]
        xp = oracle.simulate(0, g["par"], t / (K - 1), U[k], U[k + 1], X[k])
        assert np.abs(lin - xp).max() < 1e-11
        assert np.abs(lin - g["xprop"][k]).max() < 1e-11


def test_variants_fixed_time_consistency(oracle):
    """FOH + fixed final time (SCvx variant): A identical, z_fixed == z_vt + s*sigma."""
    g = np.load(os.path.join(GOLDEN, "rocketquat_dd_K15.npz"))
    X, U, t = g["X"], g["U"], float(g["t"])
    A1, B1, C1, S1, Z1 = oracle.discretize(0, g["par"], X, U, t, foh=True, vt=True)
    A2, B2, C2, _, Z2 = oracle.discretize(0, g["par"], X, U, t, foh=True, vt=False)
    assert np.abs(A1 - A2).max() < 1e-12
    assert np.abs(B1 - B2).max() < 1e-9 * np.abs(B1).max()
    assert np.abs(Z2 - (Z1 + S1 * t)).max() < 1e-11
