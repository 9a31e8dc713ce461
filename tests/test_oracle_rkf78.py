"""The restated RKF78 tableau (Boost absent): order conditions + measured convergence order."""
import numpy as np


def test_tableau_row_sums_and_quadrature_conditions(oracle):
    c, a, b = oracle.rkf78_tableau()
    assert np.abs(a.sum(axis=1) - c).max() < 1e-14  # cancellation in rows with large +- entries
    assert np.allclose(np.triu(a), 0.0)
    for k in range(8):  # sum b_i c_i^k = 1/(k+1) up to order 8
        assert abs((b * c**k).sum() - 1.0 / (k + 1)) < 1e-14, k
    # a few higher tree conditions: sum b_i a_ij c_j^k = 1/((k+1)(k+2))
    for k in range(1, 6):
        assert abs((b @ (a @ c**k)) - 1.0 / ((k + 1) * (k + 2))) < 1e-14, k
    # the 8th-order weights propagate (odeint runge_kutta_fehlberg78): b[0] = b[10] = 0, b[11] = b[12] = 41/840
    assert b[0] == 0.0 and b[10] == 0.0 and abs(b[11] - 41 / 840) < 1e-17 and abs(b[12] - 41 / 840) < 1e-17


def test_convergence_order_is_eight(oracle):
    errs = []
    for n in (8, 16, 32):
        y = oracle.rkf78_harmonic(3.0, 2.0, n)
        errs.append(abs(y[0] - np.cos(6.0)) + abs(y[1] + 3.0 * np.sin(6.0)))
    p1 = np.log2(errs[0] / errs[1])
    p2 = np.log2(errs[1] / errs[2])
    assert p1 > 7.5 and p2 > 7.8, (errs, p1, p2)  # -> 8 asymptotically
