"""Golden vectors of the THIRD model (Lander3dof, csrc/model_lander3dof.h -- not a model of the reference): f, A = df/dx, B = df/du from an independent
sympy statement of the flow map at 32 seeded random points -> lander3dof_jacobians.npz (tests/test_model_jacobian_rows.py compares the generated
analytic rows / table and forward-mode AD of the C++ plugin with it).  usage: python tests/golden/generate_lander3dof_goldens.py"""
import os

import numpy as np
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))


def lander3dof_sym():
    x = sp.symbols("x0:7")
    u = sp.symbols("u0:3")
    p = sp.symbols("p0:4")
    T = sp.Matrix(u)
    f = sp.Matrix.zeros(7, 1)
    f[0] = -p[0] * sp.sqrt(T.dot(T))
    for i in range(3):
        f[1 + i] = x[4 + i]
        f[4 + i] = u[i] / x[0] + p[1 + i]
    return x, u, p, f


if __name__ == "__main__":
    rng = np.random.default_rng(20261001)
    x, u, p, f = lander3dof_sym()
    fn = sp.lambdify([x, u, p], [f, f.jacobian(sp.Matrix(x)), f.jacobian(sp.Matrix(u))], "numpy")
    N = 32
    X = rng.uniform(-1, 1, (N, 7)); U = rng.uniform(-1, 1, (N, 3)); P = rng.uniform(0.5, 2.0, (N, 4))
    X[:, 0] = rng.uniform(0.5, 2.0, N)  # mass > 0
    U[:, 2] = rng.uniform(0.2, 1.0, N)  # |T| > 0
    F = np.zeros((N, 7)); AA = np.zeros((N, 7, 7)); BB = np.zeros((N, 7, 3))
    for i in range(N):
        fi, ai, bi = fn(X[i], U[i], P[i])
        F[i] = np.asarray(fi, dtype=float).ravel(); AA[i] = np.asarray(ai, dtype=float); BB[i] = np.asarray(bi, dtype=float)
    np.savez(os.path.join(HERE, "lander3dof_jacobians.npz"), x=X, u=U, par=P, f=F, A=AA, B=BB)
    print("lander3dof_jacobians.npz written")
