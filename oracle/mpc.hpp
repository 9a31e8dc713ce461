// ORACLE (test infrastructure, NOT product code).
//
// CPU restatement of the reference's linear MPC path:
//   scpp_core/src/discretization.cpp:9-40     exactLinearDiscretization (matrix exponential of the augmented systems)
//   scpp_core/src/MPCProblem.cpp:6-87         buildMPCProblem
//   scpp_core/src/MPCAlgorithm.cpp:11-139     MPCAlgorithm (loadParameters / initialize / setInitialState / solve)
//   scpp_models/src/rocket2d.cpp:40-84        getOperatingPoint, addApplicationConstraints
//   scpp/src/MPC_sim.cpp:16-86                closed loop
// Two solvers on the same problem:
//   kind 0 "literal"   : the reference's formulation (X, U, error_cost, input_cost; dynamics as equalities) handed to the
//                        generic ECOS restatement oracle/socp.hpp after a Ruiz equilibration (ECOS equilibrates internally);
//   kind 1 "condensed" : the states eliminated through the (constant) dynamics, 2(K-1)+2 variables, inequality-only;
//                        iterate-level twin of scpp_amd/csrc/mpc_kernel.h (same rows, same scaling, same predictor-corrector).
// Known reference defects on this path, and what is restated instead:
//   * rocket2d.cpp:43 `u << 0, -p.g_I * p.m;` streams a scalar and a 2-vector into a 2-vector (assert in debug builds,
//     out-of-bounds write otherwise).  Restated as the evident intent, the hover input (0, -g_y m).
//   * MPCProblem.cpp:67 sizes the intermediate-cost segments with v_X.cols() where rows are meant, so
//     intermediate_cost_active=true is out of bounds for K != state_dim + 1; only `false` (shipped) is restated.
//   * MPCProblem.cpp:28-31 adds the initial-state equality state_dim times; the copies are redundant and it is added once.
//   * MPC_sim.cpp:62,67 advances the plant by the measured wall time of the solve (floored at 10 ms); here the step is the
//     deterministic floor `min_timestep` = 0.010 s.
// Parity status: UNPINNED at the ECOS boundary (no ECOS binary or reference output exists here); pinned by problems with
// known optima, by literal-vs-condensed agreement, and (expm) against scipy.linalg.expm in tests/test_oracle_mpc.py.
#pragma once
#include "discretization.hpp"
#include "models.hpp"
#include "socp.hpp"
#include "structured_ipm.hpp"

namespace oracle
{

// ---------------------------------------------------------------------------------------------------------------------
// exp(A), n x n row-major: scaling and squaring with a [m/m] Pade approximant, m in {3,5,7,9,13} chosen from the 1-norm
// (N. J. Higham, "The scaling and squaring method for the matrix exponential revisited", SIAM J. Matrix Anal. Appl. 2005 --
// the algorithm behind Eigen's MatrixBase::exp() that discretization.cpp:27,37 calls).
namespace mexp
{
inline void matmul(int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
        {
            double s = 0.;
            for (int k = 0; k < n; k++)
                s += A[i * n + k] * B[k * n + j];
            C[i * n + j] = s;
        }
}
// solve P X = Q in place (Q <- X), partial pivoting
inline void luSolve(int n, std::vector<double> P, std::vector<double> &Q)
{
    for (int c = 0; c < n; c++)
    {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(P[r * n + c]) > std::fabs(P[piv * n + c]))
                piv = r;
        if (piv != c)
            for (int j = 0; j < n; j++)
            {
                std::swap(P[c * n + j], P[piv * n + j]);
                std::swap(Q[c * n + j], Q[piv * n + j]);
            }
        const double d = 1. / P[c * n + c];
        for (int r = c + 1; r < n; r++)
        {
            const double f = P[r * n + c] * d;
            if (f == 0.)
                continue;
            for (int j = c; j < n; j++)
                P[r * n + j] -= f * P[c * n + j];
            for (int j = 0; j < n; j++)
                Q[r * n + j] -= f * Q[c * n + j];
        }
    }
    for (int c = n - 1; c >= 0; c--)
        for (int j = 0; j < n; j++)
        {
            double s = Q[c * n + j];
            for (int k = c + 1; k < n; k++)
                s -= P[c * n + k] * Q[k * n + j];
            Q[c * n + j] = s / P[c * n + c];
        }
}
} // namespace mexp

inline std::vector<double> expm(int n, const std::vector<double> &Ain)
{
    using namespace mexp;
    static const double b3[] = {120., 60., 12., 1.};
    static const double b5[] = {30240., 15120., 3360., 420., 30., 1.};
    static const double b7[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
    static const double b9[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
    static const double b13[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                                 129060195264000.,   10559470521600.,    670442572800.,    33522128640.,
                                 1323241920.,        40840800.,          960960.,          16380.,
                                 182.,               1.};
    double norm1 = 0.;
    for (int j = 0; j < n; j++)
    {
        double s = 0.;
        for (int i = 0; i < n; i++)
            s += std::fabs(Ain[i * n + j]);
        norm1 = std::max(norm1, s);
    }
    const size_t nn = size_t(n) * n;
    std::vector<double> A(Ain), I(nn, 0.), U(nn), V(nn);
    for (int i = 0; i < n; i++)
        I[i * n + i] = 1.;
    int squarings = 0;
    auto padeLow = [&](const double *b, int m) {
        // U = A * sum_{odd} b_{2j+1} A^{2j},  V = sum_{even} b_{2j} A^{2j}
        std::vector<double> A2(nn), Pw(I), T(nn), Uo(nn, 0.);
        matmul(n, A.data(), A.data(), A2.data());
        std::fill(V.begin(), V.end(), 0.);
        for (int j = 0; 2 * j <= m; j++)
        {
            for (size_t e = 0; e < nn; e++)
            {
                V[e] += b[2 * j] * Pw[e];
                if (2 * j + 1 <= m)
                    Uo[e] += b[2 * j + 1] * Pw[e];
            }
            matmul(n, Pw.data(), A2.data(), T.data());
            Pw = T;
        }
        matmul(n, A.data(), Uo.data(), U.data());
    };
    if (norm1 < 1.495585217958292e-2)
        padeLow(b3, 3);
    else if (norm1 < 2.539398330063230e-1)
        padeLow(b5, 5);
    else if (norm1 < 9.504178996162932e-1)
        padeLow(b7, 7);
    else if (norm1 < 2.097847961257068e0)
        padeLow(b9, 9);
    else
    {
        const double theta13 = 5.371920351148152;
        if (norm1 > theta13)
        {
            squarings = std::max(0, int(std::ceil(std::log2(norm1 / theta13))));
            const double sc = std::ldexp(1., -squarings);
            for (auto &v : A)
                v *= sc;
        }
        std::vector<double> A2(nn), A4(nn), A6(nn), T(nn), W(nn);
        matmul(n, A.data(), A.data(), A2.data());
        matmul(n, A2.data(), A2.data(), A4.data());
        matmul(n, A4.data(), A2.data(), A6.data());
        const double *b = b13;
        for (size_t e = 0; e < nn; e++)
            T[e] = b[13] * A6[e] + b[11] * A4[e] + b[9] * A2[e];
        matmul(n, A6.data(), T.data(), W.data());
        for (size_t e = 0; e < nn; e++)
            W[e] += b[7] * A6[e] + b[5] * A4[e] + b[3] * A2[e] + b[1] * I[e];
        matmul(n, A.data(), W.data(), U.data());
        for (size_t e = 0; e < nn; e++)
            T[e] = b[12] * A6[e] + b[10] * A4[e] + b[8] * A2[e];
        matmul(n, A6.data(), T.data(), V.data());
        for (size_t e = 0; e < nn; e++)
            V[e] += b[6] * A6[e] + b[4] * A4[e] + b[2] * A2[e] + b[0] * I[e];
    }
    // (V - U) R = (V + U)
    std::vector<double> P(nn), Q(nn);
    for (size_t e = 0; e < nn; e++)
    {
        P[e] = V[e] - U[e];
        Q[e] = V[e] + U[e];
    }
    luSolve(n, P, Q);
    std::vector<double> T(nn);
    for (int s = 0; s < squarings; s++)
    {
        matmul(n, Q.data(), Q.data(), T.data());
        Q = T;
    }
    return Q;
}

// discretization.cpp:9-40
template <class Model>
void exactLinearDiscretization(const Model &model, double ts, const double *x_eq, const double *u_eq, double *A, double *B,
                               double *z)
{
    constexpr int NX = Model::NX, NU = Model::NU;
    double Ac[NX * NX], Bc[NX * NU], f[NX];
    model.computeJacobians(x_eq, u_eq, Ac, Bc);
    model.computef(x_eq, u_eq, f);
    {
        const int n = NX + NU;
        std::vector<double> E(size_t(n) * n, 0.);
        for (int i = 0; i < NX; i++)
        {
            for (int j = 0; j < NX; j++)
                E[i * n + j] = Ac[i * NX + j] * ts;
            for (int j = 0; j < NU; j++)
                E[i * n + NX + j] = Bc[i * NU + j] * ts;
        }
        const std::vector<double> X = expm(n, E);
        for (int i = 0; i < NX; i++)
        {
            for (int j = 0; j < NX; j++)
                A[i * NX + j] = X[i * n + j];
            for (int j = 0; j < NU; j++)
                B[i * NU + j] = X[i * n + NX + j];
        }
    }
    {
        const int n = NX + 1;
        std::vector<double> E(size_t(n) * n, 0.);
        for (int i = 0; i < NX; i++)
        {
            double r = f[i];
            for (int j = 0; j < NX; j++)
            {
                E[i * n + j] = Ac[i * NX + j] * ts;
                r -= Ac[i * NX + j] * x_eq[j];
            }
            for (int j = 0; j < NU; j++)
                r -= Bc[i * NU + j] * u_eq[j];
            E[i * n + NX] = r * ts;
        }
        const std::vector<double> X = expm(n, E);
        for (int i = 0; i < NX; i++)
            z[i] = X[i * n + NX];
    }
}

// rocket2d.cpp:40-44 (intent, see header)
inline void getOperatingPoint(const Rocket2d &m, double *x, double *u)
{
    for (int i = 0; i < 6; i++)
        x[i] = 0.;
    u[0] = 0.;
    u[1] = -m.p.g_I[1] * m.p.m;
}

// elimination-order keys for the sparse LDL (not part of the maths): backward Riccati order -- x_K-1 (Hessian from the
// error cone), then the dynamics rows into it, then u_K-2, then x_K-2 ... so that no pivot is the static regularisation alone
struct MPCKeys
{
    int K;
    int stageCone(int) const { return 0; }
    int xVar(int k) const { return 10 + 4 * (K - 1 - k); }
    int dyn(int k) const { return 10 + 4 * (K - 1 - (k + 1)) + 1; } // couples x_k, u_k, x_k+1
    int uVar(int k) const { return 10 + 4 * (K - 1 - (k + 1)) + 2; }
    int stageEq(int k) const { return xVar(k) + 1; }
    int globalCone() const { return 0; }
    int globalVar() const { return 1000001; }
};

// ---------------------------------------------------------------------------------------------------------------------
// Condensed problem data (instance independent when constant_dynamics): rows s = h - G v in K,
//   v = [u_0 .. u_{N-1} (NU each) | input_cost | error_cost] / D,   h = E (c0 + P x0 + Q x_final)
// row order: LP rows [per stage k=1..N: +tilt, -tilt, +rate, -rate | per input j: +gimbal, -gimbal, thrust lo, thrust hi],
// then the cones: N glide-slope cones (dim 2), the error cone (dim 1+NX), the input cone (dim 1+NU N).
struct MpcCondensed
{
    static constexpr int NX = 6, NU = 2;
    int K = 0, N = 0, nv = 0, nlp = 0, m = 0;
    std::vector<int> cone_off, cone_dim;
    std::vector<double> G, P, Q, c0; // [m][nv], [m][NX], [m][NX], [m]   (already scaled by E rows, D columns)
    std::vector<double> D, E, c;     // column scales, row scales, scaled cost
    // stage-0 checks (constants of the problem): rows s0 = c0_0 + P_0 x0 that must be in their cones
    double tan_gs, theta_max, w_max;
    // prediction matrices for getSolution: x_k = Phi_k x0 + sum_j Gam_{k,j} u_j + zeta_k
    std::vector<double> Phi, Gam, zeta; // [K][NX][NX], [K][N][NX][NU], [K][NX]
};

inline MpcCondensed buildCondensed(const Rocket2d &model, int K, const double *A, const double *B, const double *z,
                                   const double *w_term, const double *w_in, const double *x_scale_ref)
{
    constexpr int NX = 6, NU = 2;
    MpcCondensed q;
    const int N = K - 1;
    q.K = K;
    q.N = N;
    q.nv = NU * N + 2;
    q.nlp = 8 * N;
    q.tan_gs = model.p.tan_gamma_gs;
    q.theta_max = model.p.theta_max;
    q.w_max = model.p.w_B_max;
    q.Phi.assign(size_t(K) * NX * NX, 0.);
    q.Gam.assign(size_t(K) * N * NX * NU, 0.);
    q.zeta.assign(size_t(K) * NX, 0.);
    for (int i = 0; i < NX; i++)
        q.Phi[i * NX + i] = 1.;
    auto Phi = [&](int k) { return &q.Phi[size_t(k) * NX * NX]; };
    auto Gam = [&](int k, int j) { return &q.Gam[(size_t(k) * N + j) * NX * NU]; };
    auto zeta = [&](int k) { return &q.zeta[size_t(k) * NX]; };
    for (int k = 1; k < K; k++)
    {
        for (int i = 0; i < NX; i++)
        {
            for (int j = 0; j < NX; j++)
            {
                double s = 0.;
                for (int l = 0; l < NX; l++)
                    s += A[i * NX + l] * Phi(k - 1)[l * NX + j];
                Phi(k)[i * NX + j] = s;
            }
            double s = z[i];
            for (int l = 0; l < NX; l++)
                s += A[i * NX + l] * zeta(k - 1)[l];
            zeta(k)[i] = s;
        }
        for (int j = 0; j < k - 1; j++)
            for (int i = 0; i < NX; i++)
                for (int c = 0; c < NU; c++)
                {
                    double s = 0.;
                    for (int l = 0; l < NX; l++)
                        s += A[i * NX + l] * Gam(k - 1, j)[l * NU + c];
                    Gam(k, j)[i * NU + c] = s;
                }
        for (int i = 0; i < NX * NU; i++)
            Gam(k, k - 1)[i] = B[i];
    }
    // rows in unscaled form: s = cst + ax' x_k + sum_v gv[v] v + af' x_final
    struct Row
    {
        std::vector<double> g, p, qf;
        double c0;
    };
    std::vector<Row> rows;
    auto stateRow = [&](int k, const double *ax, double cst, const double *af) {
        Row r;
        r.g.assign(q.nv, 0.);
        r.p.assign(NX, 0.);
        r.qf.assign(NX, 0.);
        r.c0 = cst;
        for (int i = 0; i < NX; i++)
        {
            if (ax[i] == 0.)
                continue;
            r.c0 += ax[i] * zeta(k)[i];
            for (int j = 0; j < NX; j++)
                r.p[j] += ax[i] * Phi(k)[i * NX + j];
            for (int j = 0; j < N; j++)
                for (int c = 0; c < NU; c++)
                    r.g[j * NU + c] -= ax[i] * Gam(k, j)[i * NU + c]; // s = h - G v
        }
        if (af)
            for (int i = 0; i < NX; i++)
                r.qf[i] = af[i];
        rows.push_back(r);
    };
    auto varRow = [&](int var, double coef, double cst) {
        Row r;
        r.g.assign(q.nv, 0.);
        r.p.assign(NX, 0.);
        r.qf.assign(NX, 0.);
        r.c0 = cst;
        r.g[var] = -coef;
        rows.push_back(r);
    };
    const int v_ic = NU * N, v_ec = NU * N + 1;
    for (int k = 1; k < K; k++)
    {
        double ax[NX] = {0, 0, 0, 0, 1., 0};
        stateRow(k, ax, model.p.theta_max, nullptr);
        ax[4] = -1.;
        stateRow(k, ax, model.p.theta_max, nullptr);
        double aw[NX] = {0, 0, 0, 0, 0, 1.};
        stateRow(k, aw, model.p.w_B_max, nullptr);
        aw[5] = -1.;
        stateRow(k, aw, model.p.w_B_max, nullptr);
    }
    for (int j = 0; j < N; j++)
    {
        varRow(j * NU + 0, 1., model.p.gimbal_max);
        varRow(j * NU + 0, -1., model.p.gimbal_max);
        varRow(j * NU + 1, 1., -model.p.T_min);
        varRow(j * NU + 1, -1., model.p.T_max);
    }
    for (int k = 1; k < K; k++)
    {
        q.cone_off.push_back(int(rows.size()));
        q.cone_dim.push_back(2);
        double a1[NX] = {0, model.p.tan_gamma_gs, 0, 0, 0, 0};
        stateRow(k, a1, 0., nullptr);
        double a0[NX] = {1., 0, 0, 0, 0, 0};
        stateRow(k, a0, 0., nullptr);
    }
    q.cone_off.push_back(int(rows.size()));
    q.cone_dim.push_back(1 + NX);
    varRow(v_ec, 1., 0.);
    for (int i = 0; i < NX; i++)
    {
        double ax[NX] = {0, 0, 0, 0, 0, 0}, af[NX] = {0, 0, 0, 0, 0, 0};
        ax[i] = w_term[i];
        af[i] = -w_term[i];
        stateRow(K - 1, ax, 0., af);
    }
    q.cone_off.push_back(int(rows.size()));
    q.cone_dim.push_back(1 + NU * N);
    varRow(v_ic, 1., 0.);
    for (int j = 0; j < N; j++)
        for (int c = 0; c < NU; c++)
            varRow(j * NU + c, w_in[c], 0.);
    q.m = int(rows.size());
    // ---- scaling: columns by physical magnitudes, rows (cone-uniform) to unit max-norm ----
    q.D.assign(q.nv, 1.);
    double wtmax = 0.;
    for (int i = 0; i < NX; i++)
        wtmax = std::max(wtmax, std::fabs(w_term[i]));
    for (int j = 0; j < N; j++)
    {
        q.D[j * NU + 0] = model.p.gimbal_max;
        q.D[j * NU + 1] = model.p.T_max;
    }
    q.D[v_ic] = std::fabs(w_in[1]) * model.p.T_max;
    q.D[v_ec] = wtmax * (*x_scale_ref);
    q.E.assign(q.m, 1.);
    auto rowMax = [&](int r) {
        double mx = 0.;
        for (int v = 0; v < q.nv; v++)
            mx = std::max(mx, std::fabs(rows[r].g[v] * q.D[v]));
        return mx;
    };
    for (int r = 0; r < q.nlp; r++)
        q.E[r] = 1. / rowMax(r);
    for (size_t c = 0; c < q.cone_off.size(); c++)
    {
        double mx = 0.;
        for (int i = 0; i < q.cone_dim[c]; i++)
            mx = std::max(mx, rowMax(q.cone_off[c] + i));
        for (int i = 0; i < q.cone_dim[c]; i++)
            q.E[q.cone_off[c] + i] = 1. / mx;
    }
    q.G.assign(size_t(q.m) * q.nv, 0.);
    q.P.assign(size_t(q.m) * NX, 0.);
    q.Q.assign(size_t(q.m) * NX, 0.);
    q.c0.assign(q.m, 0.);
    for (int r = 0; r < q.m; r++)
    {
        for (int v = 0; v < q.nv; v++)
            q.G[size_t(r) * q.nv + v] = q.E[r] * rows[r].g[v] * q.D[v];
        for (int i = 0; i < NX; i++)
        {
            q.P[size_t(r) * NX + i] = q.E[r] * rows[r].p[i];
            q.Q[size_t(r) * NX + i] = q.E[r] * rows[r].qf[i];
        }
        q.c0[r] = q.E[r] * rows[r].c0;
    }
    q.c.assign(q.nv, 0.);
    const double cs = std::max(q.D[v_ic], q.D[v_ec]);
    q.c[v_ic] = q.D[v_ic] / cs;
    q.c[v_ec] = q.D[v_ec] / cs;
    return q;
}

struct MpcSolveInfo
{
    int status = 0; // 0 optimal, 1 reduced accuracy, -1 iteration limit, -2 numerics, -3 stage-0 state outside its constraints
    int iters = 0;
    double pres = 0, dres = 0, gap = 0, pcost = 0;
};

// Dense predictor-corrector on the condensed problem (no equalities): H = G' W^-2 G, one Cholesky per iteration.
class MpcCondensedIpm
{
  public:
    double feastol = 1e-8, abstol = 1e-8, reltol = 1e-8, gamma = 0.99;
    bool split_steps = true; // primal and dual step lengths of their own (csrc/mpc_kernel.h: MPC_SPLIT_STEPS); false: ECOS's common one
    int maxit = 50;
    bool verbose = false;
    explicit MpcCondensedIpm(const MpcCondensed &q_) : q(q_)
    {
        const int nv = q.nv;
        H0.assign(size_t(nv) * nv, 0.);
        for (int r = 0; r < q.m; r++)
            for (int a = 0; a < nv; a++)
                for (int b = 0; b < nv; b++)
                    H0[a * nv + b] += q.G[size_t(r) * nv + a] * q.G[size_t(r) * nv + b];
        L0 = H0;
        chol(L0);
    }
    // returns v (unscaled: inputs, input_cost, error_cost)
    MpcSolveInfo solve(const double *x0, const double *xf, std::vector<double> &v_out)
    {
        using namespace sipm;
        constexpr int NX = 6;
        const int nv = q.nv, m = q.m, nlp = q.nlp, nc = int(q.cone_off.size());
        MpcSolveInfo info;
        v_out.assign(nv, 0.);
        // stage-0 state against its own constraints
        if (!(std::fabs(x0[0]) <= q.tan_gs * x0[1]) || !(std::fabs(x0[4]) <= q.theta_max) || !(std::fabs(x0[5]) <= q.w_max))
        {
            info.status = -3;
            return info;
        }
        std::vector<double> h(m), x(nv, 0.), s(m), z(m), rz(m), rx(nv), lam(m), t(m), ds(m), dz(m), dsS(m), dzS(m), Gd(m),
            dx(nv), b(nv), w(m, 0.), eta(nc, 1.), wl(nlp, 1.), H(size_t(nv) * nv), bk_x;
        for (int r = 0; r < m; r++)
        {
            double v = q.c0[r];
            for (int i = 0; i < NX; i++)
                v += q.P[size_t(r) * NX + i] * x0[i] + q.Q[size_t(r) * NX + i] * xf[i];
            h[r] = v;
        }
        auto mulG = [&](const std::vector<double> &xv, std::vector<double> &o) {
            for (int r = 0; r < m; r++)
            {
                double a = 0.;
                for (int j = 0; j < nv; j++)
                    a += q.G[size_t(r) * nv + j] * xv[j];
                o[r] = a;
            }
        };
        auto mulGT = [&](const std::vector<double> &zv, std::vector<double> &o) {
            for (int j = 0; j < nv; j++)
            {
                double a = 0.;
                for (int r = 0; r < m; r++)
                    a += q.G[size_t(r) * nv + j] * zv[r];
                o[j] = a;
            }
        };
        auto bring2cone = [&](std::vector<double> &v) {
            // ECOS bring2cone: shift by (1 + alpha) e if outside
            double alpha = -0.99;
            for (int r = 0; r < nlp; r++)
                alpha = std::max(alpha, -v[r]);
            for (int c = 0; c < nc; c++)
            {
                const int o = q.cone_off[c], d = q.cone_dim[c];
                double n2 = 0.;
                for (int i = 1; i < d; i++)
                    n2 += v[o + i] * v[o + i];
                alpha = std::max(alpha, std::sqrt(n2) - v[o]);
            }
            const double sh = 1. + alpha;
            for (int r = 0; r < nlp; r++)
                v[r] += sh;
            for (int c = 0; c < nc; c++)
                v[q.cone_off[c]] += sh;
        };
        // ---- initial point: x = argmin ||Gx - h||, s = bring2cone(h - Gx); z = bring2cone(G x'), G'G x' = -c ----
        {
            mulGT(h, b);
            cholSolve(L0, b, x);
            mulG(x, Gd);
            for (int r = 0; r < m; r++)
                s[r] = h[r] - Gd[r];
            bring2cone(s);
            for (int j = 0; j < nv; j++)
                b[j] = -q.c[j];
            cholSolve(L0, b, dx);
            mulG(dx, z);
            bring2cone(z);
        }
        double nh = 0., ncst = 0.;
        for (int r = 0; r < m; r++)
            nh += h[r] * h[r];
        for (int j = 0; j < nv; j++)
            ncst += q.c[j] * q.c[j];
        const double resz0 = std::max(1., std::sqrt(nh)), resx0 = std::max(1., std::sqrt(ncst));
        const int Ddeg = nlp + nc;
        bool bk_valid = false;
        double pres_prev = 0.;
        for (int iter = 0;; iter++)
        {
            mulG(x, Gd);
            mulGT(z, rx);
            double gap = 0., nrz = 0., nrx = 0., nxx = 0., nzz = 0., nss = 0., pcost = 0.;
            for (int r = 0; r < m; r++)
            {
                rz[r] = s[r] + Gd[r] - h[r];
                gap += s[r] * z[r];
                nrz += rz[r] * rz[r];
                nzz += z[r] * z[r];
                nss += s[r] * s[r];
            }
            for (int j = 0; j < nv; j++)
            {
                rx[j] += q.c[j];
                nrx += rx[j] * rx[j];
                nxx += x[j] * x[j];
                pcost += q.c[j] * x[j];
            }
            const double mu = gap / Ddeg;
            const double pres = std::sqrt(nrz) / std::max(resz0 + std::sqrt(nxx) + std::sqrt(nss), 1.);
            const double dres = std::sqrt(nrx) / std::max(resx0 + std::sqrt(nzz), 1.);
            const double relgap = gap / std::max(std::fabs(pcost), 1e-300);
            info.iters = iter;
            info.pres = pres;
            info.dres = dres;
            info.gap = gap;
            info.pcost = pcost;
            if (verbose)
                std::printf("%3d pcost %+.8e gap %.2e pres %.2e dres %.2e\n", iter, pcost, gap, pres, dres);
            auto finish = [&](int st) {
                info.status = st;
                for (int j = 0; j < nv; j++)
                    v_out[j] = q.D[j] * x[j];
                return info;
            };
            // (round 6: a blown-up or out-of-cone iterate is broken with or without a fall-back -- csrc/mpc_kernel.h, csrc/ipm_solve.h IPM_BLOWN / IPM_NEG_GAP)
            if (!std::isfinite(pres) || !std::isfinite(dres) || !std::isfinite(gap) || std::fabs(gap) > 1e30 || std::fabs(pcost) > 1e30 || gap < -1e-6 ||
                (bk_valid && (pres > 500. * pres_prev || gap < 0.)))
            {
                if (!bk_valid)
                    return finish(-2);
                x = bk_x;
                return finish(1);
            }
            pres_prev = pres;
            if (pres < feastol && dres < feastol && (gap < abstol || relgap < reltol))
                return finish(0);
            const bool inacc_ok = pres < 1e-4 && dres < 1e-4 && (gap < 5e-5 || relgap < 5e-5);
            if (inacc_ok)
            {
                bk_x = x;
                bk_valid = true;
            }
            if (iter >= maxit)
                return finish(inacc_ok ? 1 : -1);
            // ---- scalings ----
            bool ok = true;
            for (int r = 0; r < nlp; r++)
            {
                if (!(s[r] > 0.) || !(z[r] > 0.))
                    ok = false;
                wl[r] = std::sqrt(s[r] / z[r]);
                lam[r] = std::sqrt(s[r] * z[r]);
            }
            std::vector<Scaling> sc(nc);
            for (int c = 0; c < nc; c++)
            {
                const int o = q.cone_off[c], d = q.cone_dim[c];
                if (!nt_scaling(&s[o], &z[o], d, sc[c]))
                    ok = false;
                else
                    applyW(sc[c], d, &z[o], &lam[o]);
            }
            if (!ok)
                return finish(inacc_ok ? 1 : -2);
            // ---- H = Gt' Gt, Gt = W^-1 G ----
            std::vector<double> Gt(size_t(m) * nv);
            for (int r = 0; r < nlp; r++)
                for (int j = 0; j < nv; j++)
                    Gt[size_t(r) * nv + j] = q.G[size_t(r) * nv + j] / wl[r];
            for (int c = 0; c < nc; c++)
            {
                const int o = q.cone_off[c], d = q.cone_dim[c];
                double col[17], out[17];
                for (int j = 0; j < nv; j++)
                {
                    for (int i = 0; i < d; i++)
                        col[i] = q.G[size_t(o + i) * nv + j];
                    applyWinv(sc[c], d, col, out);
                    for (int i = 0; i < d; i++)
                        Gt[size_t(o + i) * nv + j] = out[i];
                }
            }
            std::fill(H.begin(), H.end(), 0.);
            for (int r = 0; r < m; r++)
                for (int a = 0; a < nv; a++)
                    for (int bb = 0; bb < nv; bb++)
                        H[a * nv + bb] += Gt[size_t(r) * nv + a] * Gt[size_t(r) * nv + bb];
            std::vector<double> L(H);
            if (!chol(L))
                return finish(inacc_ok ? 1 : -2);
            double sigma_c = 0., alpha = 1., alpha_d = 1.;
            for (int pass = 0; pass < 2; pass++)
            {
                const double om = 1. - sigma_c;
                for (int r = 0; r < nlp; r++)
                {
                    const double corr = pass ? (sigma_c * mu - ds[r] * dz[r]) / s[r] : 0.;
                    t[r] = (z[r] / s[r]) * om * rz[r] - z[r] + corr;
                }
                for (int c = 0; c < nc; c++)
                {
                    const int o = q.cone_off[c], d = q.cone_dim[c];
                    double a[17], b2[17], dsv[17], u[17];
                    for (int i = 0; i < d; i++)
                        a[i] = om * rz[o + i];
                    applyWinv2(sc[c], d, a, b2);
                    if (pass == 0)
                        for (int i = 0; i < d; i++)
                            t[o + i] = b2[i] - z[o + i];
                    else
                    {
                        conicProduct(d, &dsS[o], &dzS[o], dsv);
                        for (int i = 0; i < d; i++)
                            dsv[i] = -dsv[i];
                        dsv[0] += sigma_c * mu;
                        conicDivision(d, &lam[o], dsv, u);
                        for (int i = 0; i < d; i++)
                            u[i] -= lam[o + i];
                        applyWinv(sc[c], d, u, a);
                        for (int i = 0; i < d; i++)
                            t[o + i] = b2[i] + a[i];
                    }
                }
                // H dx = -om rx - G' t ; dz = W^-2 G dx + t ; ds = -om rz - G dx
                mulGT(t, b);
                for (int j = 0; j < nv; j++)
                    b[j] = -om * rx[j] - b[j];
                cholSolve(L, b, dx);
                double chk = 0.;
                for (int j = 0; j < nv; j++)
                    chk += dx[j] * 0.;
                if (!(chk == 0.))
                    return finish(inacc_ok ? 1 : -2);
                mulG(dx, Gd);
                double ainv = 0., ainv_d = 0.; // 1 / alpha_max of the slacks' / the multipliers' direction
                for (int r = 0; r < nlp; r++)
                {
                    dz[r] = (z[r] / s[r]) * Gd[r] + t[r];
                    ds[r] = -om * rz[r] - Gd[r];
                    ainv = std::max(ainv, -ds[r] / s[r]);
                    ainv_d = std::max(ainv_d, -dz[r] / z[r]);
                }
                for (int c = 0; c < nc; c++)
                {
                    const int o = q.cone_off[c], d = q.cone_dim[c];
                    double a[17];
                    applyWinv2(sc[c], d, &Gd[o], a);
                    for (int i = 0; i < d; i++)
                    {
                        dz[o + i] = a[i] + t[o + i];
                        ds[o + i] = -om * rz[o + i] - Gd[o + i];
                    }
                    applyWinv(sc[c], d, &ds[o], &dsS[o]);
                    applyW(sc[c], d, &dz[o], &dzS[o]);
                    ainv = std::max(ainv, stepInv(d, &lam[o], &dsS[o]));
                    ainv_d = std::max(ainv_d, stepInv(d, &lam[o], &dzS[o]));
                }
                // primal and dual step lengths of their own in the corrector pass (the device kernel's MPC_SPLIT_STEPS, round 6); the centring
                // parameter follows ECOS's rule on the common affine step length
                if (!split_steps || pass == 0)
                    ainv = ainv_d = std::max(ainv, ainv_d);
                if (pass == 0)
                {
                    const double alpha_a = ainv > 0. ? std::min(1. / ainv, 1.) : 1.;
                    sigma_c = (1. - alpha_a) * (1. - alpha_a) * (1. - alpha_a);
                    sigma_c = std::min(1., std::max(1e-4, sigma_c));
                }
                else
                {
                    alpha = ainv > 0. ? std::min(gamma / ainv, 1.) : 1.;
                    alpha = std::min(alpha, 0.999);
                    alpha = std::max(alpha, 1e-8);
                    alpha_d = ainv_d > 0. ? std::min(gamma / ainv_d, 1.) : 1.;
                    alpha_d = std::max(std::min(alpha_d, 0.999), 1e-8);
                }
            }
            for (int j = 0; j < nv; j++)
                x[j] += alpha * dx[j];
            for (int r = 0; r < m; r++)
            {
                s[r] += alpha * ds[r];
                z[r] += alpha_d * dz[r];
            }
        }
    }

  private:
    const MpcCondensed &q;
    std::vector<double> H0, L0;
    bool chol(std::vector<double> &L) const
    {
        const int n = q.nv;
        for (int j = 0; j < n; j++)
        {
            double d = L[j * n + j];
            for (int k = 0; k < j; k++)
                d -= L[j * n + k] * L[j * n + k];
            if (!(d > 0.) || !std::isfinite(d))
                return false;
            d = std::sqrt(d);
            L[j * n + j] = d;
            for (int i = j + 1; i < n; i++)
            {
                double v = L[i * n + j];
                for (int k = 0; k < j; k++)
                    v -= L[i * n + k] * L[j * n + k];
                L[i * n + j] = v / d;
            }
        }
        return true;
    }
    void cholSolve(const std::vector<double> &L, const std::vector<double> &b, std::vector<double> &x) const
    {
        const int n = q.nv;
        x = b;
        for (int i = 0; i < n; i++)
        {
            double v = x[i];
            for (int k = 0; k < i; k++)
                v -= L[i * n + k] * x[k];
            x[i] = v / L[i * n + i];
        }
        for (int i = n - 1; i >= 0; i--)
        {
            double v = x[i];
            for (int k = i + 1; k < n; k++)
                v -= L[k * n + i] * x[k];
            x[i] = v / L[i * n + i];
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// MPCAlgorithm.cpp:11-139
class MPCAlgorithm
{
  public:
    static constexpr int NX = 6, NU = 2;
    std::shared_ptr<Rocket2d> model;
    size_t K = 0;
    bool nondimensionalize = false, constant_dynamics = true, intermediate_cost_active = false;
    double time_horizon = 0.;
    double state_weights_intermediate[NX], state_weights_terminal[NX], input_weights[NU];
    double A[NX * NX], B[NX * NU], z[NX];
    double x_init[NX], x_final[NX];
    std::vector<double> X, U; // [K][NX], [K-1][NU]
    int solver_kind = 1;      // 0 literal, 1 condensed
    MpcSolveInfo last;
    double error_cost = 0., input_cost = 0.;

    MPCAlgorithm(std::shared_ptr<Rocket2d> m, const std::string &folder) : model(m)
    {
        // MPCAlgorithm.cpp:17-32
        ParameterServer param(folder + "/MPC.info");
        param.loadScalar("K", K);
        param.loadScalar("nondimensionalize", nondimensionalize);
        param.loadScalar("constant_dynamics", constant_dynamics);
        param.loadScalar("intermediate_cost_active", intermediate_cost_active);
        param.loadScalar("time_horizon", time_horizon);
        param.loadVector("state_weights_intermediate", state_weights_intermediate, NX);
        param.loadVector("state_weights_terminal", state_weights_terminal, NX);
        param.loadVector("input_weights", input_weights, NU);
    }
    // MPCAlgorithm.cpp:34-69
    void initialize()
    {
        if (nondimensionalize || intermediate_cost_active || model->p.constrain_initial_final)
            throw std::runtime_error("oracle MPC: only the shipped mode (dimensional, terminal cost; constant_dynamics is immaterial, "
                                     "constrain_initial_final=false) is restated");
        TrajectoryData dummy;
        model->getNewModelParameters(dummy);
        double x_eq[NX], u_eq[NU];
        getOperatingPoint(*model, x_eq, u_eq);
        const double dt = time_horizon / double(K - 1);
        exactLinearDiscretization(*model, dt, x_eq, u_eq, A, B, z);
        X.assign(K * NX, 0.);
        U.assign((K - 1) * NU, 0.);
        const double xref = std::sqrt(model->p.x_init[0] * model->p.x_init[0] + model->p.x_init[1] * model->p.x_init[1]);
        cond = buildCondensed(*model, int(K), A, B, z, state_weights_terminal, input_weights, &xref);
        ipm.reset(new MpcCondensedIpm(cond));
    }
    void setInitialState(const double *x) { std::copy(x, x + NX, x_init); }
    void setFinalState(const double *x) { std::copy(x, x + NX, x_final); }
    void setTolerances(double feastol, double abstol, double reltol, int maxit)
    {
        ipm->feastol = feastol;
        ipm->abstol = abstol;
        ipm->reltol = reltol;
        ipm->maxit = maxit;
    }
    // MPCAlgorithm.cpp:95-139 ; returns the solver status
    int solve()
    {
        if (solver_kind == 1)
        {
            std::vector<double> v;
            last = ipm->solve(x_init, x_final, v);
            if (last.status >= 0)
            {
                const int N = int(K) - 1;
                for (int i = 0; i < NU * N; i++)
                    U[i] = v[i];
                input_cost = v[NU * N];
                error_cost = v[NU * N + 1];
                for (size_t k = 0; k < K; k++)
                    for (int i = 0; i < NX; i++)
                    {
                        double a = cond.zeta[k * NX + i];
                        for (int j = 0; j < NX; j++)
                            a += cond.Phi[(k * NX + i) * NX + j] * x_init[j];
                        for (int j = 0; j < N; j++)
                            for (int c = 0; c < NU; c++)
                                a += cond.Gam[((k * N + j) * NX + i) * NU + c] * U[j * NU + c];
                        X[k * NX + i] = a;
                    }
            }
            return last.status;
        }
        return solveLiteral();
    }
    const MpcCondensed &condensed() const { return cond; }

  private:
    MpcCondensed cond;
    std::unique_ptr<MpcCondensedIpm> ipm;

    // MPCProblem.cpp:6-87 + rocket2d.cpp:46-84, equilibrated, solved by oracle/socp.hpp
    int solveLiteral()
    {
        const int Kk = int(K), N = Kk - 1;
        MPCKeys key{Kk};
        Socp socp;
        const int iX = socp.addVars(NX * Kk, 0), iU = socp.addVars(NU * N, 0);
        auto vX = [&](int i, int k) { return iX + k * NX + i; };
        auto vU = [&](int i, int k) { return iU + k * NU + i; };
        for (int k = 0; k < Kk; k++)
            for (int i = 0; i < NX; i++)
                socp.setKey(vX(i, k), key.xVar(k));
        for (int k = 0; k < N; k++)
            for (int i = 0; i < NU; i++)
                socp.setKey(vU(i, k), key.uVar(k));
        const int v_ec = socp.addVars(1, key.globalVar()), v_ic = socp.addVars(1, key.globalVar());
        for (int i = 0; i < NX; i++)
            socp.addEq(Aff(-x_init[i]).add(vX(i, 0), 1.), key.stageEq(0));
        for (int k = 0; k < N; k++)
            for (int i = 0; i < NX; i++)
            {
                Aff e(z[i]);
                for (int j = 0; j < NX; j++)
                    e.add(vX(j, k), A[i * NX + j]);
                for (int j = 0; j < NU; j++)
                    e.add(vU(j, k), B[i * NU + j]);
                e.add(vX(i, k + 1), -1.);
                socp.addEq(e, key.dyn(k));
            }
        {
            std::vector<Aff> e;
            e.push_back(Aff().add(v_ec, 1.));
            for (int i = 0; i < NX; i++)
                e.push_back(Aff(-state_weights_terminal[i] * x_final[i]).add(vX(i, Kk - 1), state_weights_terminal[i]));
            socp.addSoc(e, key.globalCone());
            socp.c[v_ec] = 1.;
        }
        {
            std::vector<Aff> e;
            e.push_back(Aff().add(v_ic, 1.));
            for (int k = 0; k < N; k++)
                for (int i = 0; i < NU; i++)
                    e.push_back(Aff().add(vU(i, k), input_weights[i]));
            socp.addSoc(e, key.globalCone());
            socp.c[v_ic] = 1.;
        }
        model->addApplicationConstraints(socp, Kk, N, vX, vU, key);
        // ---- Ruiz equilibration (ECOS equilibrates inside its setup): x = Dc xt ----
        std::vector<double> Dc(socp.n, 1.);
        auto scaleRows = [&](std::vector<SocpRow> &rows, bool uniform) {
            double mx_all = 0.;
            std::vector<double> mx(rows.size(), 0.);
            for (size_t r = 0; r < rows.size(); r++)
            {
                for (auto &p : rows[r].t)
                    mx[r] = std::max(mx[r], std::fabs(p.second));
                mx_all = std::max(mx_all, mx[r]);
            }
            for (size_t r = 0; r < rows.size(); r++)
            {
                const double e = 1. / std::sqrt(std::max(uniform ? mx_all : mx[r], 1e-300));
                for (auto &p : rows[r].t)
                    p.second *= e;
                rows[r].rhs *= e;
            }
        };
        for (int pass = 0; pass < 4; pass++)
        {
            std::vector<double> cm(socp.n, 0.);
            auto colMax = [&](const std::vector<SocpRow> &rows) {
                for (auto &r : rows)
                    for (auto &p : r.t)
                        cm[p.first] = std::max(cm[p.first], std::fabs(p.second));
            };
            colMax(socp.eq);
            colMax(socp.lp);
            for (auto &cn : socp.soc)
                colMax(cn);
            for (int j = 0; j < socp.n; j++)
                cm[j] = 1. / std::sqrt(std::max(cm[j], 1e-300));
            auto colScale = [&](std::vector<SocpRow> &rows) {
                for (auto &r : rows)
                    for (auto &p : r.t)
                        p.second *= cm[p.first];
            };
            colScale(socp.eq);
            colScale(socp.lp);
            for (auto &cn : socp.soc)
                colScale(cn);
            for (int j = 0; j < socp.n; j++)
            {
                Dc[j] *= cm[j];
                socp.c[j] *= cm[j];
            }
            scaleRows(socp.eq, false);
            scaleRows(socp.lp, false);
            for (auto &cn : socp.soc)
                scaleRows(cn, true);
        }
        double cmax = 0.;
        for (double v : socp.c)
            cmax = std::max(cmax, std::fabs(v));
        for (double &v : socp.c)
            v /= cmax;
        SocpSolver solver(socp);
        solver.opt.verbose = std::getenv("ORACLE_MPC_VERBOSE") != nullptr;
        SocpResult r = solver.solve();
        last.status = r.exitflag == 0 ? 0 : r.exitflag == 10 ? 1 : r.exitflag == 1 ? -3 : (r.exitflag < 0 ? r.exitflag : -2);
        last.iters = r.iter;
        last.pres = r.pres;
        last.dres = r.dres;
        last.gap = r.gap;
        last.pcost = r.pcost;
        if (r.exitflag == 0 || r.exitflag == 10)
        {
            for (int k = 0; k < Kk; k++)
                for (int i = 0; i < NX; i++)
                    X[size_t(k) * NX + i] = Dc[vX(i, k)] * r.x[vX(i, k)];
            for (int k = 0; k < N; k++)
                for (int i = 0; i < NU; i++)
                    U[size_t(k) * NU + i] = Dc[vU(i, k)] * r.x[vU(i, k)];
            error_cost = Dc[v_ec] * r.x[v_ec];
            input_cost = Dc[v_ic] * r.x[v_ic];
        }
        return last.status;
    }
};

// MPC_sim.cpp:16-86 with the deterministic plant step (see header).  A failed solve holds the previous input.
struct MPCSimResult
{
    std::vector<double> X_sim, U_sim, t_sim; // per executed step
    int steps = 0, failed_solves = 0, ipm_iters = 0;
    bool reached = false;
};

inline MPCSimResult runMPCSim(MPCAlgorithm &solver, const double *x_start, double sim_time = 15., double min_timestep = 0.010,
                              int max_steps = 1 << 30)
{
    constexpr int NX = 6, NU = 2;
    MPCSimResult res;
    Rocket2d &model = *solver.model;
    double x[NX], u[NU] = {0., 0.};
    std::copy(x_start, x_start + NX, x);
    solver.setFinalState(model.p.x_final);
    double t = 0.;
    while (t < sim_time && res.steps < max_steps)
    {
        solver.setInitialState(x);
        const int st = solver.solve();
        res.ipm_iters += solver.last.iters;
        simulate(model, min_timestep, u, u, x);
        t += min_timestep;
        if (st >= 0)
        {
            u[0] = solver.U[0];
            u[1] = solver.U[1];
        }
        else
            res.failed_solves++;
        res.X_sim.insert(res.X_sim.end(), x, x + NX);
        res.U_sim.insert(res.U_sim.end(), u, u + NU);
        res.t_sim.push_back(t);
        res.steps++;
        double d2 = 0.;
        for (int i = 0; i < NX; i++)
            d2 += (x[i] - model.p.x_final[i]) * (x[i] - model.p.x_final[i]);
        if (std::sqrt(d2) < 0.02)
        {
            res.reached = true;
            break;
        }
    }
    return res;
}

} // namespace oracle
