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


@pytest.mark.parametrize("name,foh,vt", [("foh_fixed", True, False), ("zoh_vt", False, True), ("zoh_fixed", False, False)])
def test_dd_variants_match_dop853_goldens(oracle, name, foh, vt):
    """The three other variants of multipleShootingImplementation<FOH, VT> (discretization.cpp:42-55) -- first-order hold with a
    fixed final time is what SCvx (the headline mode) and SC with free_final_time false discretise with -- against DOP853 goldens
    generated without the oracle (tests/golden/generate_dd_variant_goldens.py)."""
    g = np.load(os.path.join(GOLDEN, "rocketquat_dd_variants_K15.npz"))
    U = g["U"] if foh else g["U"][:-1]
    out = oracle.discretize(0, g["par"], g["X"], U, float(g["t"]), foh=foh, vt=vt)
    for n, a in zip("ABCSZ", out):
        o = g[f"{name}_{n}"]
        if (n == "C" and not foh) or (n == "S" and not vt):
            assert not o.any() and (a.size == 0 or not a.any()), n  # empty in the reference (discretizationData.hpp:56-65)
            continue
        assert np.abs(a - o).max() <= 1e-10 * max(1.0, np.abs(o).max()), (name, n)
