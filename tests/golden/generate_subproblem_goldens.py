"""Independent pin of the RocketQuat SC and SCvx SUB-PROBLEMS.

Nothing here imports the oracle, the HIP library or scpp_amd: the scenario, the nondimensionalisation, the initial-guess
trajectory, the discretisation and the two convex sub-problems are restated from the reference text in numpy / sympy and
solved with a general-purpose scipy optimiser (trust-constr on the epigraph form with concave cone functions), so that the
optimum recorded in rocketquat_subproblem_K5.npz owes nothing to the interior-point solvers it is used to check.

  scenario        scpp_models/config/RocketQuat/model.info:109-213 (the active "FALCON 9" block), SC.info, SCvx.info
  loading/scaling scpp_models/src/rocketQuat.cpp:234-311, scpp_models/include/common.hpp:30-38
  initial guess   scpp_models/src/rocketQuat.cpp:39-68  (alpha2 = k/K: the reference's own interpolation)
  discretisation  scpp_core/include/discretizationImplementation.hpp:38-181 -- here as forward sensitivities integrated by
                  DOP853 (rtol 1e-13), the formulation generate_goldens.py already uses for G2
  SC sub-problem  scpp_core/src/SCProblem.cpp:6-138 + scpp_models/src/rocketQuat.cpp:70-144
  SCvx sub-problem scpp_core/src/SCvxProblem.cpp:6-71 + the same model constraints

K = 5 keeps the NLP at ~210 variables.  Run: python tests/golden/generate_subproblem_goldens.py
"""
import os
import sys

import numpy as np
import sympy as sp
from scipy.integrate import solve_ivp
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from generate_goldens import rocketquat_sym  # the symbolic flow map only (pure sympy)

K = 5
NX, NU = 14, 4
DEG = np.pi / 180.0


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def euler_xyz(rpy):  # common.hpp:30-38: AngleAxis(x, X) * AngleAxis(y, Y) * AngleAxis(z, Z)
    r, p, y = rpy
    qx = np.array([np.cos(r / 2), np.sin(r / 2), 0, 0])
    qy = np.array([np.cos(p / 2), 0, np.sin(p / 2), 0])
    qz = np.array([np.cos(y / 2), 0, 0, np.sin(y / 2)])
    return quat_mul(quat_mul(qx, qy), qz)


def scenario(x_init_dimensional=None):
    """model.info:109-213 -> Parameters::loadFromFile -> nondimensionalize (rocketQuat.cpp:234-311).  x_init_dimensional (SI units, [m, r, v, q, w]):
    another initial state of the same vehicle (the randomised instances of the bench), scaled by ITS mass and distance (rocketQuat.cpp:293-294)"""
    g_I = np.array([0.0, 0.0, -9.81]); J_B = np.array([5e6, 5e6, 7e4]); r_T_B = np.array([0.0, 0.0, -15.0])
    m_init, m_dry = 24000.0, 22000.0
    r_init = np.array([200.0, 200.0, 800.0]); v_init = np.array([-40.0, -40.0, -80.0])
    rpy_init = np.array([-20.0, 20.0, 0.0]) * DEG
    I_sp, T_min, T_max = 275.0, 200000.0, 420000.0
    gimbal_max, theta_max, gamma_gs, w_B_max = 15 * DEG, 90 * DEG, 30 * DEG, 60 * DEG
    final_time = 12.0
    alpha_m = 1.0 / (I_sp * abs(g_I[2]))
    x_init = np.concatenate([[m_init], r_init, v_init, euler_xyz(rpy_init), np.zeros(3)])
    if x_init_dimensional is not None:
        x_init = np.array(x_init_dimensional, dtype=float).copy()
    x_final = np.concatenate([[m_dry], np.zeros(3), np.zeros(3), euler_xyz(np.zeros(3)), np.zeros(3)])
    ms, rs = x_init[0], np.linalg.norm(x_init[1:4])
    alpha_m *= rs; r_T_B = r_T_B / rs; g_I = g_I / rs; J_B = J_B / (ms * rs * rs)
    x_init[0] /= ms; x_init[1:7] /= rs
    x_final[0] /= ms; x_final[1:7] /= rs
    T_min /= ms * rs; T_max /= ms * rs
    par = np.concatenate([[alpha_m], g_I, J_B, r_T_B])  # flow-map parameter vector (rocketQuat.cpp:146-154)
    return dict(par=par, x_init=x_init, x_final=x_final, T_min=T_min, T_max=T_max, final_time=final_time,
                gs=np.tan(gamma_gs), tilt=np.sqrt((1 - np.cos(theta_max)) / 2), wmax=w_B_max, gim=np.tan(gimbal_max),
                m_scale=ms, r_scale=rs)


def initial_trajectory(sc):
    X = np.zeros((K, NX)); U = np.zeros((K, NU))
    x0, xf = sc["x_init"], sc["x_final"]
    for k in range(K):
        a1, a2 = (K - k) / K, k / K
        X[k, 0:7] = a1 * x0[0:7] + a2 * xf[0:7]
        q0, q1 = x0[7:11], xf[7:11]
        d = float(q0 @ q1)
        if abs(d) >= 1 - np.finfo(float).eps:  # Eigen::Quaternion::slerp
            s0, s1 = 1 - a2, a2
        else:
            th = np.arccos(abs(d))
            s0, s1 = np.sin((1 - a2) * th) / np.sin(th), np.sin(a2 * th) / np.sin(th)
        if d < 0:
            s1 = -s1
        X[k, 7:11] = s0 * q0 + s1 * q1
        X[k, 11:14] = a1 * x0[11:14] + a2 * xf[11:14]
        U[k] = [0, 0, (sc["T_max"] - sc["T_min"]) / 2, 0]
    return X, U, sc["final_time"]


def discretize(sc, X, U, sigma, variable_time):
    x_, u_, p_, f_ = rocketquat_sym()
    fn = sp.lambdify([x_, u_, p_], [f_, f_.jacobian(sp.Matrix(x_)), f_.jacobian(sp.Matrix(u_))], "numpy")
    par = sc["par"]
    dt = 1.0 / (K - 1) if variable_time else sigma / (K - 1)
    scale = sigma if variable_time else 1.0
    A = np.zeros((K - 1, NX, NX)); B = np.zeros((K - 1, NX, NU)); C = np.zeros((K - 1, NX, NU)); S = np.zeros((K - 1, NX)); Z = np.zeros((K - 1, NX))
    for k in range(K - 1):
        def rhs(tau, y):
            x = y[:14]; Phi = y[14:210].reshape(14, 14); PB = y[210:266].reshape(14, 4); PC = y[266:322].reshape(14, 4); ps = y[322:336]; pz = y[336:350]
            u = U[k] + tau / dt * (U[k + 1] - U[k])
            fx, a, b = fn(x, u, par)
            fx = np.asarray(fx, dtype=float).ravel(); a = np.asarray(a, dtype=float) * scale; b = np.asarray(b, dtype=float) * scale
            zdot = a @ pz - a @ x - b @ u + (0.0 if variable_time else fx)  # :109 / :115
            return np.concatenate([scale * fx, (a @ Phi).ravel(), (a @ PB + b * (dt - tau) / dt).ravel(), (a @ PC + b * tau / dt).ravel(), a @ ps + fx, zdot])
        y0 = np.concatenate([X[k], np.eye(14).ravel(), np.zeros(56 + 56 + 14 + 14)])
        y = solve_ivp(rhs, [0, dt], y0, method="DOP853", rtol=1e-13, atol=1e-16).y[:, -1]
        A[k] = y[14:210].reshape(14, 14); B[k] = y[210:266].reshape(14, 4); C[k] = y[266:322].reshape(14, 4); S[k] = y[322:336]; Z[k] = y[336:350]
    if not variable_time:
        S[:] = 0.0
    return A, B, C, S, Z


FINAL_FIXED = [1, 2, 3, 4, 5, 6, 8, 9, 11, 12, 13]


class SubProblem:
    """Smooth NLP form.  v = [X (K*14) | U (K*4) | nu+ | nu- ((K-1)*14 each) | delta (K) | sigma | delta_sigma] (SC) or
    v = [X | U | nu+ | nu-] (SCvx).  Second-order cones t >= ||w|| enter as t^2 - ||w||^2 >= 0, t >= 0."""

    def __init__(self, sc, Xb, Ub, sb, dd, mode, w):
        self.sc, self.Xb, self.Ub, self.sb, self.dd, self.mode, self.w = sc, Xb, Ub, sb, dd, mode, w
        self.nX, self.nU, self.nN = K * NX, K * NU, (K - 1) * NX
        self.oU = self.nX; self.oP = self.oU + self.nU; self.oM = self.oP + self.nN
        self.oD = self.oM + self.nN
        self.n = self.oD + (K + 2 if mode == "sc" else 0)

    def split(self, v):
        X = v[:self.nX].reshape(K, NX); U = v[self.oU:self.oP].reshape(K, NU)
        P = v[self.oP:self.oM].reshape(K - 1, NX); M = v[self.oM:self.oD].reshape(K - 1, NX)
        if self.mode == "sc":
            return X, U, P, M, v[self.oD:self.oD + K], v[self.oD + K], v[self.oD + K + 1]
        return X, U, P, M, None, self.sb, 0.0

    def cost(self, v):
        X, U, P, M, D, sig, dsg = self.split(v)
        c = self.w["vc"] * (P.sum() + M.sum())
        if self.mode == "sc":
            c += self.w["t"] * sig + self.w["trt"] * dsg + self.w["trx"] * D.sum()
        return c

    def cost_grad(self, v):
        g = np.zeros(self.n)
        g[self.oP:self.oD] = self.w["vc"]
        if self.mode == "sc":
            g[self.oD:self.oD + K] = self.w["trx"]; g[self.oD + K] = self.w["t"]; g[self.oD + K + 1] = self.w["trt"]
        return g

    def eq(self, v):
        X, U, P, M, D, sig, dsg = self.split(v)
        A, B, C, S, Z = self.dd
        # (the reference states some of these twice -- X(13,0) through x_init and through X.row(13) = 0, U(3,K-1) through the final
        #  input and through U.row(3) = 0; each is kept once so that the equality Jacobian has full row rank)
        r = [X[0] - self.sc["x_init"], X[K - 1][FINAL_FIXED] - self.sc["x_final"][FINAL_FIXED], U[K - 1][[0, 1, 3]], X[1:K - 1, 13], U[:K - 1, 3]]
        for k in range(K - 1):
            r.append(X[k + 1] - (A[k] @ X[k] + B[k] @ U[k] + C[k] @ U[k + 1] + S[k] * sig + Z[k] + P[k] - M[k]))
        return np.concatenate(r)

    # ---- inequalities >= 0: linear rows and squared second-order cones t^2 - ||w||^2 >= 0, each entry a sparse linear form
    #      const + sum coef * v[idx]; values and the exact Jacobian come from the same description ----
    def _forms(self):
        if hasattr(self, "_lin"):
            return self._lin, self._soc
        sc = self.sc
        iX = lambda k, j: k * NX + j
        iU = lambda k, j: self.oU + k * NU + j
        lin, soc = [], []  # lin: (const, [(idx, coef)]) ; soc: (t_form, [w_forms])
        for i in range(self.nN):
            lin.append((0.0, [(self.oP + i, 1.0)]))
        for i in range(self.nN):
            lin.append((0.0, [(self.oM + i, 1.0)]))
        for k in range(K):
            lin.append((-sc["x_final"][0], [(iX(k, 0), 1.0)]))  # mass >= m_dry
        free = range(1, K - 1)  # nodes 0 and K-1 have r, q_xy, w fixed by the equalities: their cones are constants
        for k in free:
            soc.append(((0.0, [(iX(k, 3), sc["gs"])]), [(0.0, [(iX(k, 1), 1.0)]), (0.0, [(iX(k, 2), 1.0)])]))
            lin.append((0.0, [(iX(k, 3), 1.0)]))
            soc.append(((sc["tilt"], []), [(0.0, [(iX(k, 8), 1.0)]), (0.0, [(iX(k, 9), 1.0)])]))
            soc.append(((sc["wmax"], []), [(0.0, [(iX(k, 11 + j), 1.0)]) for j in range(3)]))
        for k in range(K):
            lin.append((-sc["T_min"], [(iU(k, 2), 1.0)]))  # linearised minimum thrust with thrust_const = (0, 0, 1) (rocketQuat.cpp:113-121,156-173)
            soc.append(((sc["T_max"], []), [(0.0, [(iU(k, j), 1.0)]) for j in range(3)]))
            soc.append(((0.0, [(iU(k, 2), sc["gim"])]), [(0.0, [(iU(k, 0), 1.0)]), (0.0, [(iU(k, 1), 1.0)])]))
        if self.mode == "sc":
            oD = self.oD
            lin.append((-0.001, [(oD + K, 1.0)]))  # sigma >= 0.001
            # (sigma - sigma0)^2 <= delta_sigma  as  delta_sigma - (sigma - sigma0)^2 >= 0: a "cone" with t^2 replaced by a linear term
            self._dsg_row = (oD + K + 1, (-self.sb, [(oD + K, 1.0)]))
            for k in range(K):
                w = [(-self.Xb[k, j], [(iX(k, j), 1.0)]) for j in range(NX)] + [(-self.Ub[k, j], [(iU(k, j), 1.0)]) for j in range(NU)]
                soc.append(((0.0, [(oD + k, 1.0)]), w))
                lin.append((0.0, [(oD + k, 1.0)]))
        else:
            for k in range(K):
                soc.append(((self.w["tr"], []), [(-self.Ub[k, j], [(iU(k, j), 1.0)]) for j in range(NU)]))
        self._lin, self._soc = lin, soc
        return lin, soc

    @staticmethod
    def _val(form, v):
        c, terms = form
        return c + sum(cf * v[i] for i, cf in terms)

    EPS = 1e-12  # smoothing of ||w|| at w = 0 (a trust-region cone whose node does not move)

    def ineq(self, v):  # >= 0 ; cones as the CONCAVE functions t - ||w||: every SQP linearisation is an outer approximation
        lin, soc = self._forms()
        r = [self._val(f, v) for f in lin]
        for t, ws in soc:
            r.append(self._val(t, v) - np.sqrt(sum(self._val(w, v) ** 2 for w in ws) + self.EPS ** 2))
        if self.mode == "sc":
            idx, f = self._dsg_row
            r.append(v[idx] - self._val(f, v) ** 2)
        return np.array(r)

    def ineq_jac(self, v):
        lin, soc = self._forms()
        n_rows = len(lin) + len(soc) + (1 if self.mode == "sc" else 0)
        J = np.zeros((n_rows, self.n))
        r = 0
        for c, terms in lin:
            for i, cf in terms:
                J[r, i] += cf
            r += 1
        for t, ws in soc:
            for i, cf in t[1]:
                J[r, i] += cf
            wvs = [self._val(w, v) for w in ws]
            nrm = np.sqrt(sum(x * x for x in wvs) + self.EPS ** 2)
            for w, wv in zip(ws, wvs):
                for i, cf in w[1]:
                    J[r, i] -= wv / nrm * cf
            r += 1
        if self.mode == "sc":
            idx, f = self._dsg_row
            J[r, idx] += 1.0
            fv = self._val(f, v)
            for i, cf in f[1]:
                J[r, i] -= 2 * fv * cf
        return J

    def eq_jac(self, v):
        if not hasattr(self, "_Je"):
            self._Je = _jac(self.eq, np.zeros(self.n))  # the equalities are linear: constant Jacobian
        return self._Je

    def start(self):
        """the linearisation point itself with the dynamics defect absorbed by the virtual control: strictly inside the cones"""
        A, B, C, S, Z = self.dd
        X, U = self.Xb.copy(), self.Ub.copy()
        X[0] = self.sc["x_init"]; X[K - 1][FINAL_FIXED] = self.sc["x_final"][FINAL_FIXED]; X[:, 13] = 0; U[:, 3] = 0; U[K - 1][[0, 1]] = 0
        P = np.zeros((K - 1, NX)); M = np.zeros((K - 1, NX))
        for k in range(K - 1):
            d = X[k + 1] - (A[k] @ X[k] + B[k] @ U[k] + C[k] @ U[k + 1] + S[k] * self.sb + Z[k])
            P[k] = np.maximum(d, 0) + 1e-3; M[k] = np.maximum(-d, 0) + 1e-3
        v = np.concatenate([X.ravel(), U.ravel(), P.ravel(), M.ravel()])
        if self.mode == "sc":
            dn = np.sqrt(((X - self.Xb) ** 2).sum(1) + ((U - self.Ub) ** 2).sum(1))
            v = np.concatenate([v, dn + 1e-2, [self.sb, 1e-2]])
        return v

    def solve(self):
        """scipy trust-constr (an interior-point / SQP trust-region method for general NLPs) from the cold start, then a
        second run from its own solution with the barrier restarted: the objective must not move."""
        cons = [{"type": "eq", "fun": self.eq, "jac": self.eq_jac}, {"type": "ineq", "fun": self.ineq, "jac": self.ineq_jac}]
        v0 = self.start()
        assert np.abs(self.ineq_jac(v0) - _jac(self.ineq, v0)).max() < 1e-5  # analytic cone gradients vs central differences (smooth point)
        opts = dict(maxiter=4000, gtol=1e-11, xtol=1e-14, barrier_tol=1e-11, initial_barrier_parameter=0.1)
        r = minimize(self.cost, self.start(), jac=self.cost_grad, method="trust-constr", constraints=cons, options=opts)
        print("  trust-constr pass 0:", r.message if hasattr(r, "message") else "", "iterations", r.nit, "obj %.12f" % r.fun, "constr viol %.2e" % r.constr_violation, flush=True)
        r2 = minimize(self.cost, r.x, jac=self.cost_grad, method="trust-constr", constraints=cons,
                      options=dict(opts, initial_barrier_parameter=1e-6, initial_tr_radius=1e-2))
        print("  trust-constr pass 1: iterations", r2.nit, "obj %.12f" % r2.fun, "constr viol %.2e" % r2.constr_violation, flush=True)
        return (r2.x, r2) if r2.constr_violation <= max(r.constr_violation, 1e-9) and abs(r2.fun - r.fun) < 1e-4 else (r.x, r)


def _jac(f, v):
    """central differences (exact for the linear equalities; a check of the analytic cone gradients elsewhere)"""
    f0 = f(v)
    J = np.zeros((f0.size, v.size))
    h = 1e-6
    for j in range(v.size):
        e = np.zeros(v.size); e[j] = h
        J[:, j] = (f(v + e) - f(v - e)) / (2 * h)
    return J


def kkt_report(pb, v):
    """first-order optimality of the recorded point, independent of the optimiser that produced it: least-squares
    multipliers of the active set, then stationarity and sign conditions"""
    g = pb.cost_grad(v)
    Je, Ji, ci = pb.eq_jac(v), pb.ineq_jac(v), pb.ineq(v)
    act = ci < 1e-7
    Jn = np.vstack([Je, Ji[act]])
    lam = np.linalg.lstsq(Jn.T, g, rcond=None)[0]
    stat = np.abs(Jn.T @ lam - g).max()
    mu = lam[Je.shape[0]:]
    return dict(stationarity=float(stat), min_ineq_multiplier=float(mu.min()) if mu.size else 0.0, max_eq_violation=float(np.abs(pb.eq(v)).max()),
                min_ineq=float(ci.min()), n_active=int(act.sum()))


def main():
    sc = scenario()
    Xb, Ub, sb = initial_trajectory(sc)
    out = dict(par=sc["par"], x_init=sc["x_init"], x_final=sc["x_final"], Xbar=Xb, Ubar=Ub, sigma_bar=sb,
               consts=np.array([sc["T_min"], sc["T_max"], sc["gs"], sc["tilt"], sc["wmax"], sc["gim"], sc["m_scale"], sc["r_scale"]]))
    # ---- SC sub-problem (free final time) with the shipped SC.info weights ----
    w_sc = dict(t=1.0, trt=1.0, trx=50.0, vc=1000.0)
    dd = discretize(sc, Xb, Ub, sb, True)
    pb = SubProblem(sc, Xb, Ub, sb, dd, "sc", w_sc)
    v, r = pb.solve()
    X, U, P, M, D, sig, dsg = pb.split(v)
    rep = kkt_report(pb, v)
    print("SC  : obj %.12f" % r.fun, rep)
    for n, a in zip("ABCSZ", dd):
        out["sc_" + n] = a
    out.update(sc_X=X.copy(), sc_U=U.copy(), sc_nu=(P - M).copy(), sc_sigma=sig, sc_delta=D.copy(), sc_delta_sigma=dsg, sc_objective=r.fun,
               sc_norm1_nu=float((P + M).sum()), sc_weights=np.array([1.0, 1.0, 50.0, 1000.0]),
               sc_kkt=np.array([rep["stationarity"], rep["min_ineq_multiplier"], rep["max_eq_violation"], rep["min_ineq"]]))
    # ---- SCvx sub-problem (fixed final time, hard input trust region) with the shipped SCvx.info ----
    w_vx = dict(vc=1000.0, tr=5.0)
    ddv = discretize(sc, Xb, Ub, sb, False)
    pv = SubProblem(sc, Xb, Ub, sb, ddv, "scvx", w_vx)
    vv, rv = pv.solve()
    Xv, Uv, Pv, Mv, _, _, _ = pv.split(vv)
    repv = kkt_report(pv, vv)
    print("SCvx: obj %.12f" % rv.fun, repv)
    for n, a in zip("ABCSZ", ddv):
        out["scvx_" + n] = a
    out.update(scvx_X=Xv.copy(), scvx_U=Uv.copy(), scvx_nu=(Pv - Mv).copy(), scvx_objective=rv.fun, 
               scvx_norm1_nu=float((Pv + Mv).sum()), scvx_weights=np.array([1000.0, 5.0]),
               scvx_kkt=np.array([repv["stationarity"], repv["min_ineq_multiplier"], repv["max_eq_violation"], repv["min_ineq"]]))
    np.savez(os.path.join(HERE, "rocketquat_subproblem_K5.npz"), **out)
    print("written rocketquat_subproblem_K5.npz")


if __name__ == "__main__":
    main()
