"""DOP853 goldens for the three OTHER variants of multipleShootingImplementation<INTERPOLATE_INPUT, VARIABLE_TIME>
(scpp_core/include/discretizationImplementation.hpp:38-181, dispatcher discretization.cpp:42-55): first-order hold with a fixed
final time (what SCvxAlgorithm and SCAlgorithm with free_final_time false discretise with), zero-order hold with variable and with
fixed final time.  G2 (generate_goldens.py: dd_golden) pins the shipped <true, true> variant; this file completes the set.

Independent of the oracle and of the kernels: the linearisation point (X, U, t, par) is read from rocketquat_dd_K15.npz, the flow map
and its Jacobians are sympy's (generate_goldens.py: rocketquat_sym), the matrices come from forward sensitivities
    Phi' = a Phi,  PB' = a PB + b w_B(tau),  PC' = a PC + b w_C(tau),  ps' = a ps + f,  pz' = a pz + (c - a x - b u)
integrated by scipy DOP853 (rtol 1e-13) -- algebraically what the reference's Phi^-1 formulation produces (A = Phi(dt), B = PB(dt), ...):
  variable time (:58-62,:106-109): a = sigma df/dx, b = sigma df/du, dt = 1/(K-1), c = 0, ps drives S;
  fixed time    (:112-115):        a = df/dx, b = df/du, dt = t/(K-1), c = f, no S;
  first-order hold (:87-93): u(tau) = u_k + tau/dt (u_{k+1} - u_k), w_B = (dt - tau)/dt, w_C = tau/dt;
  zero-order hold  (:96-101): u(tau) = u_k, w_B = 1, no C.
    python tests/golden/generate_dd_variant_goldens.py      -> rocketquat_dd_variants_K15.npz
"""
import os
import sys

import numpy as np
import sympy as sp
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from generate_goldens import rocketquat_sym


def variant(fn, par, X, U, t, foh, vt):
    K, nx, nu = X.shape[0], 14, 4
    dt = 1.0 / (K - 1) if vt else t / (K - 1)
    scale = t if vt else 1.0
    A = np.zeros((K - 1, nx, nx)); B = np.zeros((K - 1, nx, nu)); C = np.zeros((K - 1, nx, nu)); S = np.zeros((K - 1, nx)); Z = np.zeros((K - 1, nx))
    for k in range(K - 1):
        def rhs(tau, y):
            x = y[:14]; Phi = y[14:210].reshape(14, 14); PB = y[210:266].reshape(14, 4); PC = y[266:322].reshape(14, 4); ps = y[322:336]; pz = y[336:350]
            u = U[k] + tau / dt * (U[k + 1] - U[k]) if foh else U[k]
            fx, a, b = fn(x, u, par)
            fx = np.asarray(fx, dtype=float).ravel(); a = np.asarray(a, dtype=float) * scale; b = np.asarray(b, dtype=float) * scale
            wB, wC = ((dt - tau) / dt, tau / dt) if foh else (1.0, 0.0)
            zdot = a @ pz - a @ x - b @ u + (0.0 if vt else fx)
            return np.concatenate([scale * fx, (a @ Phi).ravel(), (a @ PB + b * wB).ravel(), (a @ PC + b * wC).ravel(), a @ ps + fx, zdot])
        y0 = np.concatenate([X[k], np.eye(14).ravel(), np.zeros(56 + 56 + 14 + 14)])
        y = solve_ivp(rhs, [0, dt], y0, method="DOP853", rtol=1e-13, atol=1e-16).y[:, -1]
        A[k] = y[14:210].reshape(14, 14); B[k] = y[210:266].reshape(14, 4); C[k] = y[266:322].reshape(14, 4); S[k] = y[322:336]; Z[k] = y[336:350]
    if not foh:
        C[:] = 0.0
    if not vt:
        S[:] = 0.0
    return A, B, C, S, Z


def main():
    g = np.load(os.path.join(HERE, "rocketquat_dd_K15.npz"))
    X, U, t, par = g["X"], g["U"], float(g["t"]), g["par"]
    x_, u_, p_, f_ = rocketquat_sym()
    fn = sp.lambdify([x_, u_, p_], [f_, f_.jacobian(sp.Matrix(x_)), f_.jacobian(sp.Matrix(u_))], "numpy")
    out = dict(X=X, U=U, t=t, par=par)
    # sanity: the <true, true> variant recomputed here reproduces G2
    A, B, C, S, Z = variant(fn, par, X, U, t, True, True)
    for n, a in zip("ABCSZ", (A, B, C, S, Z)):
        assert np.abs(a - g[n]).max() <= 1e-10 * max(1.0, np.abs(g[n]).max()), n
    for name, foh, vt in (("foh_fixed", True, False), ("zoh_vt", False, True), ("zoh_fixed", False, False)):
        for n, a in zip("ABCSZ", variant(fn, par, X, U, t, foh, vt)):
            out[f"{name}_{n}"] = a
        print(name, "done")
    np.savez(os.path.join(HERE, "rocketquat_dd_variants_K15.npz"), **out)
    print("written rocketquat_dd_variants_K15.npz")


if __name__ == "__main__":
    main()
