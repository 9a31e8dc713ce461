"""Generates the committed golden fixtures (run in the dev container; needs sympy + scipy, NOT the reference:
the reference has no tests/golden vectors of its own and cannot be built or imported here).

G1  rocketquat_jacobians.npz : f, df/dx, df/du of the RocketQuat flow map (rocketQuat.cpp:7-37 restated
    symbolically, incl. the un-normalised rotation matrix and the w x w == 0 quirk) at 32 seeded points.
    rocket2d_jacobians.npz   : same for Rocket2d (rocket2d.cpp:7-38).
G2  rocketquat_dd_K{15,50}.npz : A,B,C,s,z of the shipped Falcon-9 scenario's initial-guess trajectory,
    integrated independently with scipy DOP853 (rtol=1e-13) on the forward-sensitivity form.
G5  sc_regression.json : self-generated regression record of the oracle's SC runs (labelled as such).
"""
import json
import os
import sys

import numpy as np
import sympy as sp
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def rocketquat_sym():
    x = sp.symbols("x0:14")
    u = sp.symbols("u0:4")
    p = sp.symbols("p0:10")
    m = x[0]
    v = sp.Matrix(x[4:7])
    qw, qx, qy, qz = x[7:11]
    w = sp.Matrix(x[11:14])
    T = sp.Matrix(u[0:3])
    R = sp.Matrix([
        [1 - 2 * (qy**2 + qz**2), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
        [2 * (qx * qy + qw * qz), 1 - 2 * (qx**2 + qz**2), 2 * (qy * qz - qw * qx)],
        [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx**2 + qy**2)],
    ])
    Om = sp.Matrix([[0, -w[0], -w[1], -w[2]], [w[0], 0, w[2], -w[1]], [w[1], -w[2], 0, w[0]], [w[2], w[1], -w[0], 0]])
    q = sp.Matrix([qw, qx, qy, qz])
    g = sp.Matrix(p[1:4])
    Jinv = sp.diag(1 / p[4], 1 / p[5], 1 / p[6])
    rT = sp.Matrix(p[7:10])
    f = sp.Matrix.zeros(14, 1)
    f[0] = -p[0] * sp.sqrt(T.dot(T))
    f[1:4, 0] = v
    f[4:7, 0] = R * T / m + g
    f[7:11, 0] = sp.Rational(1, 2) * Om * q
    f[11:14, 0] = Jinv * (rT.cross(T) + sp.Matrix([0, 0, u[3]])) - w.cross(w)
    return x, u, p, f


def rocket2d_sym():
    x = sp.symbols("x0:6")
    u = sp.symbols("u0:2")
    p = sp.symbols("p0:6")
    TB = sp.Matrix([[sp.cos(u[0]), -sp.sin(u[0])], [sp.sin(u[0]), sp.cos(u[0])]]) * sp.Matrix([0, u[1]])
    Re = sp.Matrix([[sp.cos(x[4]), -sp.sin(x[4])], [sp.sin(x[4]), sp.cos(x[4])]])
    f = sp.Matrix.zeros(6, 1)
    f[0] = x[2]
    f[1] = x[3]
    acc = Re * TB / p[0] + sp.Matrix([p[2], p[3]])
    f[2] = acc[0]
    f[3] = acc[1]
    f[4] = x[5]
    f[5] = (p[4] * TB[1] - p[5] * TB[0]) / p[1]
    return x, u, p, f


def jac_golden(symf, nx, nu, npar, name, rng):
    x, u, p, f = symf()
    A = f.jacobian(sp.Matrix(x))
    Bm = f.jacobian(sp.Matrix(u))
    fn = sp.lambdify([x, u, p], [f, A, Bm], "numpy")
    N = 32
    X = rng.uniform(-1, 1, (N, nx))
    U = rng.uniform(-1, 1, (N, nu))
    P = rng.uniform(0.5, 2.0, (N, npar))
    if name == "rocketquat":
        X[:, 0] = rng.uniform(0.5, 2.0, N)  # mass > 0
        U[:, 2] = rng.uniform(0.2, 1.0, N)  # |T| > 0
    F = np.zeros((N, nx)); AA = np.zeros((N, nx, nx)); BB = np.zeros((N, nx, nu))
    for i in range(N):
        fi, ai, bi = fn(X[i], U[i], P[i])
        F[i] = np.asarray(fi, dtype=float).ravel(); AA[i] = np.asarray(ai, dtype=float); BB[i] = np.asarray(bi, dtype=float)
    np.savez(os.path.join(HERE, f"{name}_jacobians.npz"), x=X, u=U, par=P, f=F, A=AA, B=BB)


def dd_golden(K):
    import oracle_lib as O
    import scpp_amd

    m = scpp_amd.RocketQuat().loadParameters()
    sc = O.SC(O.ROCKETQUAT, K=K)
    sc.set_solver(1)
    sc.solve()
    X, U, t = sc.iterate(0)  # nondimensional initial guess (rocketQuat.cpp:39-68)
    par = m.flow_params()
    nx, nu = 14, 4
    dt = 1.0 / (K - 1)
    A = np.zeros((K - 1, nx, nx)); Bm = np.zeros((K - 1, nx, nu)); C = np.zeros((K - 1, nx, nu)); S = np.zeros((K - 1, nx)); Z = np.zeros((K - 1, nx))
    xprop = np.zeros((K - 1, nx))
    x_, u_, p_, f_ = rocketquat_sym()
    fn = sp.lambdify([x_, u_, p_], [f_, f_.jacobian(sp.Matrix(x_)), f_.jacobian(sp.Matrix(u_))], "numpy")
    for k in range(K - 1):
        def rhs(tau, y):
            x = y[:14]; Phi = y[14:210].reshape(14, 14); PB = y[210:266].reshape(14, 4); PC = y[266:322].reshape(14, 4); ps = y[322:336]; pz = y[336:350]
            u = U[k] + tau / dt * (U[k + 1] - U[k])
            fx, a, b = fn(x, u, par)
            fx = np.asarray(fx, dtype=float).ravel(); a = np.asarray(a, dtype=float) * t; b = np.asarray(b, dtype=float) * t
            return np.concatenate([t * fx, (a @ Phi).ravel(), (a @ PB + b * (dt - tau) / dt).ravel(), (a @ PC + b * tau / dt).ravel(), a @ ps + fx, a @ pz - a @ x - b @ u])
        y0 = np.concatenate([X[k], np.eye(14).ravel(), np.zeros(56 + 56 + 14 + 14)])
        sol = solve_ivp(rhs, [0, dt], y0, method="DOP853", rtol=1e-13, atol=1e-16)
        y = sol.y[:, -1]
        xprop[k] = y[:14]; A[k] = y[14:210].reshape(14, 14); Bm[k] = y[210:266].reshape(14, 4); C[k] = y[266:322].reshape(14, 4); S[k] = y[322:336]; Z[k] = y[336:350]
    np.savez(os.path.join(HERE, f"rocketquat_dd_K{K}.npz"), X=X, U=U, t=t, par=par, A=A, B=Bm, C=C, S=S, Z=Z, xprop=xprop)


def sc_regression():
    import oracle_lib as O

    rec = {"note": "SELF-GENERATED regression record of the oracle (the reference publishes no outputs); not a reference golden."}
    sc = O.SC(O.ROCKET2D)
    sc.solve()
    inf = sc.info()
    rec["rocket2d_K30_literal"] = {"iterations": sc.meta()["iterations"], "converged": sc.meta()["converged"], "sigma": float(inf[-1, 3]),
                                   "norm1_nu": inf[:, 0].tolist(), "sum_delta": inf[:, 1].tolist()}
    for K in (15, 50):
        sc = O.SC(O.ROCKETQUAT, K=K)
        sc.set_solver(1)
        sc.solve()
        inf = sc.info()
        rec[f"rocketquat_K{K}_structured"] = {"iterations": sc.meta()["iterations"], "converged": sc.meta()["converged"], "sigma": inf[:, 3].tolist(),
                                              "norm1_nu": inf[:, 0].tolist(), "sum_delta": inf[:, 1].tolist(), "ipm_iters": inf[:, 4].tolist()}
    json.dump(rec, open(os.path.join(HERE, "sc_regression.json"), "w"), indent=1)


if __name__ == "__main__":
    rng = np.random.default_rng(20260927)
    jac_golden(rocketquat_sym, 14, 4, 10, "rocketquat", rng)
    jac_golden(rocket2d_sym, 6, 2, 6, "rocket2d", rng)
    dd_golden(15)
    dd_golden(50)
    sc_regression()
    print("goldens written")
