"""Independent pin of the ROCKET2D sub-problems: the first SC and SCvx sub-problem of the shipped Rocket2D scenario solved WITHOUT any
interior-point code of this repository -- Kelley's cutting planes over HiGHS' dual simplex (the solver part of
generate_subproblem_cut_goldens.py; see its header) on problem data restated here from the reference text in numpy / sympy.  Nothing
in this file imports oracle/, scpp_amd or the HIP library, nor the RocketQuat restatement's problem class.

  scenario         scpp_models/config/Rocket2D/model.info:1-71, SC.info, SCvx.info
  loading/scaling  scpp_models/src/rocket2d.cpp:150-232 (deg2rad of the angles and rates; r_scale = ||r_init||, m_scale = m)
  flow map         scpp_models/src/rocket2d.cpp:7-38   (T_B = Rot(gimbal) (0, T); v' = Rot(eta) T_B / m + g; w' = (r_T x T_B) / J)
  initial guess    scpp_models/src/rocket2d.cpp:120-135 (alpha2 = k / K; U_k = (0, (T_max + T_min) / 2); t = final_time)
  model rows       scpp_models/src/rocket2d.cpp:46-84   (x_init, x_final, gimbal(K-1) = 0; glide slope |r_x| <= tan(gamma) r_y;
                                                          boxes on eta, w, gimbal, thrust)
  discretisation   scpp_core/include/discretizationImplementation.hpp:38-181 as forward sensitivities integrated by DOP853 (rtol 1e-13)
  SC sub-problem   scpp_core/src/SCProblem.cpp:6-138  (weights of Rocket2D/SC.info: 1, 1, 1, 1000; nondimensionalised as shipped)
  SCvx sub-problem scpp_core/src/SCvxProblem.cpp:6-71 (w_vc 1000, radius 5; SI units as shipped AND nondimensionalised)

Cases: SC at K = 25 (the reference's SC.info) and K = 30 (this repository's), SCvx at K = 30 in SI units (as shipped) and
nondimensionalised.  Run: python tests/golden/generate_rocket2d_cut_goldens.py  (about a minute) -> rocket2d_subproblem_cuts.npz
"""
import os
import sys

import numpy as np
import sympy as sp
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from generate_subproblem_cut_goldens import solve_cuts  # the LP / cutting-plane loop only

NX, NU = 6, 2
DEG = np.pi / 180.0


def scenario(nondim):
    g = np.array([0.0, -9.81]); J = 5e6; rT = np.array([0.0, -15.0]); m = 24000.0
    x_init = np.array([-200.0, 800.0, 0.0, -100.0, -20.0 * DEG, 0.0 * DEG])
    x_final = np.array([0.0, 0.0, 0.0, -1.0, 0.0, 0.0])
    T_min, T_max = 10000.0, 420000.0
    gim, gam, th, wmax = 15 * DEG, 45 * DEG, 60 * DEG, 20 * DEG
    rs, ms = 1.0, 1.0
    if nondim:  # rocket2d.cpp:200-216
        rs, ms = float(np.linalg.norm(x_init[:2])), m
        m = m / ms; rT = rT / rs; g = g / rs; J = J / (ms * rs * rs)
        x_init = x_init.copy(); x_final = x_final.copy()
        x_init[:4] /= rs; x_final[:4] /= rs
        T_min /= ms * rs; T_max /= ms * rs
    return dict(par=np.array([m, J, g[0], g[1], rT[0], rT[1]]), x_init=x_init, x_final=x_final, T_min=T_min, T_max=T_max, gim=gim,
                tan_gs=np.tan(gam), th=th, wmax=wmax, final_time=12.0, r_scale=rs, m_scale=ms)


def flow_map():
    x = sp.symbols("x0:6"); u = sp.symbols("u0:2"); p = sp.symbols("p0:6")
    m, J, gx, gy, rx, ry = p
    eta, w = x[4], x[5]
    ang, mag = u
    TB = sp.Matrix([[sp.cos(ang), -sp.sin(ang)], [sp.sin(ang), sp.cos(ang)]]) * sp.Matrix([0, mag])
    R = sp.Matrix([[sp.cos(eta), -sp.sin(eta)], [sp.sin(eta), sp.cos(eta)]])
    acc = R * TB / m + sp.Matrix([gx, gy])
    f = sp.Matrix([x[2], x[3], acc[0], acc[1], w, (rx * TB[1] - ry * TB[0]) / J])
    return sp.lambdify([x, u, p], [f, f.jacobian(sp.Matrix(x)), f.jacobian(sp.Matrix(u))], "numpy")


def initial_trajectory(sc, K):
    X = np.array([(K - k) / K * sc["x_init"] + k / K * sc["x_final"] for k in range(K)])
    U = np.tile([0.0, (sc["T_max"] + sc["T_min"]) / 2], (K, 1))
    return X, U, sc["final_time"]


def discretize(sc, K, X, U, sigma, variable_time):
    fn = flow_map()
    par = sc["par"]
    dt = 1.0 / (K - 1) if variable_time else sigma / (K - 1)
    scale = sigma if variable_time else 1.0
    nB = NX * NU
    A = np.zeros((K - 1, NX, NX)); B = np.zeros((K - 1, NX, NU)); C = np.zeros((K - 1, NX, NU)); S = np.zeros((K - 1, NX)); Z = np.zeros((K - 1, NX))
    o = [NX, NX + NX * NX, NX + NX * NX + nB, NX + NX * NX + 2 * nB, NX + NX * NX + 2 * nB + NX]
    for k in range(K - 1):
        def rhs(tau, y):
            x = y[:NX]; Phi = y[o[0]:o[1]].reshape(NX, NX); PB = y[o[1]:o[2]].reshape(NX, NU); PC = y[o[2]:o[3]].reshape(NX, NU)
            ps = y[o[3]:o[4]]; pz = y[o[4]:]
            u = U[k] + tau / dt * (U[k + 1] - U[k])
            fx, a, b = fn(x, u, par)
            fx = np.asarray(fx, dtype=float).ravel(); a = np.asarray(a, dtype=float) * scale; b = np.asarray(b, dtype=float) * scale
            zdot = a @ pz - a @ x - b @ u + (0.0 if variable_time else fx)
            return np.concatenate([scale * fx, (a @ Phi).ravel(), (a @ PB + b * (dt - tau) / dt).ravel(), (a @ PC + b * tau / dt).ravel(), a @ ps + fx, zdot])
        y0 = np.concatenate([X[k], np.eye(NX).ravel(), np.zeros(2 * nB + 2 * NX)])
        y = solve_ivp(rhs, [0, dt], y0, method="DOP853", rtol=1e-13, atol=1e-16).y[:, -1]
        A[k] = y[o[0]:o[1]].reshape(NX, NX); B[k] = y[o[1]:o[2]].reshape(NX, NU); C[k] = y[o[2]:o[3]].reshape(NX, NU); S[k] = y[o[3]:o[4]]; Z[k] = y[o[4]:]
    if not variable_time:
        S[:] = 0.0
    return A, B, C, S, Z


class SubProblem:
    """v = [X (K*6) | U (K*2) | nu+ | nu- ((K-1)*6 each) | delta (K) | sigma | delta_sigma] (SC) or [X | U | nu+ | nu-] (SCvx): the
    interface linear_parts / solve_cuts of generate_subproblem_cut_goldens.py expect (n, eq, cost_grad, _forms, mode, _dsg_row)."""

    def __init__(self, sc, K, Xb, Ub, sb, dd, mode, w):
        self.sc, self.K, self.Xb, self.Ub, self.sb, self.dd, self.mode, self.w = sc, K, Xb, Ub, sb, dd, mode, w
        self.nX, self.nU, self.nN = K * NX, K * NU, (K - 1) * NX
        self.oU = self.nX; self.oP = self.oU + self.nU; self.oM = self.oP + self.nN; self.oD = self.oM + self.nN
        self.n = self.oD + (K + 2 if mode == "sc" else 0)

    def split(self, v):
        K = self.K
        X = v[:self.nX].reshape(K, NX); U = v[self.oU:self.oP].reshape(K, NU)
        P = v[self.oP:self.oM].reshape(K - 1, NX); M = v[self.oM:self.oD].reshape(K - 1, NX)
        if self.mode == "sc":
            return X, U, P, M, v[self.oD:self.oD + K], v[self.oD + K], v[self.oD + K + 1]
        return X, U, P, M, None, self.sb, 0.0

    def cost_grad(self, v):
        g = np.zeros(self.n)
        g[self.oP:self.oD] = self.w["vc"]
        if self.mode == "sc":
            K = self.K
            g[self.oD:self.oD + K] = self.w["trx"]; g[self.oD + K] = self.w["t"]; g[self.oD + K + 1] = self.w["trt"]
        return g

    def eq(self, v):
        K = self.K
        X, U, P, M, D, sig, dsg = self.split(v)
        A, B, C, S, Z = self.dd
        r = [X[0] - self.sc["x_init"], X[K - 1] - self.sc["x_final"], U[K - 1][[0]]]  # rocket2d.cpp:55-60
        for k in range(K - 1):
            r.append(X[k + 1] - (A[k] @ X[k] + B[k] @ U[k] + C[k] @ U[k + 1] + S[k] * sig + Z[k] + P[k] - M[k]))
        return np.concatenate(r)

    def _forms(self):
        if hasattr(self, "_lin"):
            return self._lin, self._soc
        sc, K = self.sc, self.K
        iX = lambda k, j: k * NX + j  # noqa: E731
        iU = lambda k, j: self.oU + k * NU + j  # noqa: E731
        lin, soc = [], []  # lin: const + sum coef v >= 0 ; soc: t >= ||w||
        for i in range(2 * self.nN):
            lin.append((0.0, [(self.oP + i, 1.0)]))  # nu+, nu- >= 0
        for k in range(K):
            # glide slope (a norm of ONE entry): tan(gamma) r_y >= |r_x|
            lin.append((0.0, [(iX(k, 1), sc["tan_gs"]), (iX(k, 0), -1.0)]))
            lin.append((0.0, [(iX(k, 1), sc["tan_gs"]), (iX(k, 0), 1.0)]))
            for j, bnd in ((4, sc["th"]), (5, sc["wmax"])):
                lin.append((bnd, [(iX(k, j), -1.0)])); lin.append((bnd, [(iX(k, j), 1.0)]))
            lin.append((sc["gim"], [(iU(k, 0), -1.0)])); lin.append((sc["gim"], [(iU(k, 0), 1.0)]))
            lin.append((-sc["T_min"], [(iU(k, 1), 1.0)])); lin.append((sc["T_max"], [(iU(k, 1), -1.0)]))
        if self.mode == "sc":
            oD = self.oD
            lin.append((-0.001, [(oD + K, 1.0)]))  # sigma >= 0.001 (SCProblem.cpp:34)
            self._dsg_row = (oD + K + 1, (-self.sb, [(oD + K, 1.0)]))  # (sigma - sigma0)^2 <= delta_sigma (:91-96)
            for k in range(K):  # || (x - x0, u - u0) || <= delta_k (:103-125)
                w = [(-self.Xb[k, j], [(iX(k, j), 1.0)]) for j in range(NX)] + [(-self.Ub[k, j], [(iU(k, j), 1.0)]) for j in range(NU)]
                soc.append(((0.0, [(oD + k, 1.0)]), w))
        else:
            for k in range(K):  # || u - u0 || <= r (SCvxProblem.cpp:58-68)
                soc.append(((self.w["tr"], []), [(-self.Ub[k, j], [(iU(k, j), 1.0)]) for j in range(NU)]))
        self._lin, self._soc = lin, soc
        return lin, soc


    # ---- evaluation of a given point (used by tests/test_independent_path_audit.py) ----
    def cost(self, v):
        return float(self.cost_grad(v) @ v)

    def ineq(self, v):
        """every linear row and every cone t - ||w|| (and delta_sigma - (sigma - sigma0)^2 in SC mode): >= 0 at a feasible point"""
        lin, soc = self._forms()
        val = lambda f: f[0] + sum(cf * v[i] for i, cf in f[1])  # noqa: E731
        r = [val(f) for f in lin]
        for t, ws in soc:
            r.append(val(t) - np.sqrt(sum(val(w) ** 2 for w in ws)))
        if self.mode == "sc":
            idx, f = self._dsg_row
            r.append(v[idx] - val(f) ** 2)
        return np.array(r)


def solve_case(K, mode, nondim):
    sc = scenario(nondim)
    Xb, Ub, sb = initial_trajectory(sc, K)
    dd = discretize(sc, K, Xb, Ub, sb, mode == "sc")
    w = dict(t=1.0, trt=1.0, trx=1.0, vc=1000.0) if mode == "sc" else dict(vc=1000.0, tr=5.0)
    pb = SubProblem(sc, K, Xb, Ub, sb, dd, mode, w)
    print(" Rocket2D K=%d %s %s: %d variables" % (K, mode, "nondimensionalised" if nondim else "SI units", pb.n), flush=True)
    x, info = solve_cuts(pb, verbose=False)
    X, U, P, M, Dl, sig, dsg = pb.split(x)
    print("   objective %.12f  ||nu||_1 %.12f  %s" % (info["objective"], float((P + M).sum()), info), flush=True)
    out = {"X": X.copy(), "U": U.copy(), "objective": info["objective"], "norm1_nu": float((P + M).sum()), "sigma": float(sig), "sigma_bar": float(sb),
           "rounds": info["rounds"], "violations": np.array([info["cone_violation"], info["eq_violation"], info["lin_violation"]]),
           "scales": np.array([sc["r_scale"], sc["m_scale"]])}
    if mode == "sc":
        out["sum_delta"] = float(Dl.sum()); out["delta_sigma"] = float(dsg)
    return out


def main():
    out = {}
    for name, K, mode, nondim in (("sc_K25", 25, "sc", True), ("sc_K30", 30, "sc", True), ("scvx_K30_si", 30, "scvx", False), ("scvx_K30_nd", 30, "scvx", True)):
        for k, v in solve_case(K, mode, nondim).items():
            out["%s_%s" % (name, k)] = v
    np.savez(os.path.join(HERE, "rocket2d_subproblem_cuts.npz"), **out)
    print("written rocket2d_subproblem_cuts.npz")


if __name__ == "__main__":
    main()
