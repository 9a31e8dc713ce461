// Host side of the linear MPC path: what MPCAlgorithm::initialize does ONCE per controller
// (scpp_core/src/MPCAlgorithm.cpp:34-69): linearise the model at the operating point, discretise it exactly
// (scpp_core/src/discretization.cpp:9-40, a matrix exponential of the augmented system), and -- because the dynamics are
// then constant (MPC.info: constant_dynamics true) -- eliminate the states so that every later solve on the device
// (mpc_kernel.h) is a 16-variable inequality-only cone program.  O(1) work, not on the hot path.
#pragma once
#include "model_rocketquat.h"
#include "mpc_kernel.h"
#include "../../include/scpp_hip.h"
#include <cmath>
#include <cstring>

namespace scpp
{
namespace mpc
{

// dense n x n helpers (row-major, n <= 9)
struct SmallMat
{
    static constexpr int MAXN = 9;
    int n;
    double a[MAXN * MAXN];
    explicit SmallMat(int n_) : n(n_) { std::memset(a, 0, sizeof a); }
    double &operator()(int i, int j) { return a[i * n + j]; }
    double operator()(int i, int j) const { return a[i * n + j]; }
    static SmallMat identity(int n)
    {
        SmallMat m(n);
        for (int i = 0; i < n; i++)
            m(i, i) = 1.;
        return m;
    }
};
inline SmallMat operator*(const SmallMat &x, const SmallMat &y)
{
    SmallMat r(x.n);
    for (int i = 0; i < x.n; i++)
        for (int k = 0; k < x.n; k++)
        {
            const double v = x(i, k);
            if (v == 0.)
                continue;
            for (int j = 0; j < x.n; j++)
                r(i, j) += v * y(k, j);
        }
    return r;
}
inline SmallMat axpby(double al, const SmallMat &x, double be, const SmallMat &y)
{
    SmallMat r(x.n);
    for (int i = 0; i < x.n * x.n; i++)
        r.a[i] = al * x.a[i] + be * y.a[i];
    return r;
}

// exp(M): [13/13] Pade approximant after scaling ||M||_1 below Higham's theta_13, then repeated squaring
// (the top branch of the algorithm behind Eigen's .exp(); lower-degree approximants for tiny norms differ from it only in
// the last bits)
inline SmallMat expm(SmallMat M)
{
    static const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                                 129060195264000.,   10559470521600.,    670442572800.,    33522128640.,
                                 1323241920.,        40840800.,          960960.,          16380.,
                                 182.,               1.};
    const int n = M.n;
    double norm1 = 0.;
    for (int j = 0; j < n; j++)
    {
        double s = 0.;
        for (int i = 0; i < n; i++)
            s += std::fabs(M(i, j));
        norm1 = std::fmax(norm1, s);
    }
    int sq = 0;
    while (std::ldexp(norm1, -sq) > 5.371920351148152)
        sq++;
    for (int i = 0; i < n * n; i++)
        M.a[i] = std::ldexp(M.a[i], -sq);
    const SmallMat I = SmallMat::identity(n), M2 = M * M, M4 = M2 * M2, M6 = M4 * M2;
    auto comb = [&](double c6, double c4, double c2, double c0) {
        SmallMat r(n);
        for (int i = 0; i < n * n; i++)
            r.a[i] = c6 * M6.a[i] + c4 * M4.a[i] + c2 * M2.a[i] + c0 * I.a[i];
        return r;
    };
    const SmallMat U = M * axpby(1., M6 * comb(b[13], b[11], b[9], 0.), 1., comb(b[7], b[5], b[3], b[1]));
    const SmallMat V = axpby(1., M6 * comb(b[12], b[10], b[8], 0.), 1., comb(b[6], b[4], b[2], b[0]));
    // (V - U) R = V + U by Gaussian elimination with row pivoting
    SmallMat Pm = axpby(1., V, -1., U), R = axpby(1., V, 1., U);
    for (int c = 0; c < n; c++)
    {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(Pm(r, c)) > std::fabs(Pm(piv, c)))
                piv = r;
        for (int j = 0; j < n && piv != c; j++)
        {
            std::swap(Pm(c, j), Pm(piv, j));
            std::swap(R(c, j), R(piv, j));
        }
        for (int r = 0; r < n; r++)
        {
            if (r == c)
                continue;
            const double f = Pm(r, c) / Pm(c, c);
            for (int j = 0; j < n; j++)
            {
                Pm(r, j) -= f * Pm(c, j);
                R(r, j) -= f * R(c, j);
            }
        }
    }
    for (int r = 0; r < n; r++)
        for (int j = 0; j < n; j++)
            R(r, j) /= Pm(r, r);
    for (int s = 0; s < sq; s++)
        R = R * R;
    return R;
}

// discretization.cpp:9-40 ; A [NX][NX], B [NX][NU], z [NX] row-major
template <class Model>
void exactLinearDiscretization(const double *par, double ts, const double *x_eq, const double *u_eq, double *A, double *B,
                               double *z)
{
    constexpr int NXm = Model::NX, NUm = Model::NU;
    double Ac[NXm][NXm], Bc[NXm][NUm], f[NXm];
    Model::template systemFlowMap<double>(x_eq, u_eq, par, f);
    for (int d = 0; d < NXm + NUm; d++) // one forward-mode pass per input direction
    {
        Dual1 xd[NXm], ud[NUm], fd[NXm];
        for (int i = 0; i < NXm; i++)
            xd[i] = Dual1(x_eq[i], d == i ? 1. : 0.);
        for (int i = 0; i < NUm; i++)
            ud[i] = Dual1(u_eq[i], d == NXm + i ? 1. : 0.);
        Model::template systemFlowMap<Dual1>(xd, ud, par, fd);
        for (int i = 0; i < NXm; i++)
            (d < NXm ? Ac[i][d] : Bc[i][d - NXm]) = fd[i].d;
    }
    SmallMat E(NXm + NUm);
    for (int i = 0; i < NXm; i++)
    {
        for (int j = 0; j < NXm; j++)
            E(i, j) = Ac[i][j] * ts;
        for (int j = 0; j < NUm; j++)
            E(i, NXm + j) = Bc[i][j] * ts;
    }
    const SmallMat X1 = expm(E);
    for (int i = 0; i < NXm; i++)
    {
        for (int j = 0; j < NXm; j++)
            A[i * NXm + j] = X1(i, j);
        for (int j = 0; j < NUm; j++)
            B[i * NUm + j] = X1(i, NXm + j);
    }
    SmallMat E2(NXm + 1);
    for (int i = 0; i < NXm; i++)
    {
        double r = f[i];
        for (int j = 0; j < NXm; j++)
        {
            E2(i, j) = Ac[i][j] * ts;
            r -= Ac[i][j] * x_eq[j];
        }
        for (int j = 0; j < NUm; j++)
            r -= Bc[i][j] * u_eq[j];
        E2(i, NXm) = r * ts;
    }
    const SmallMat X2 = expm(E2);
    for (int i = 0; i < NXm; i++)
        z[i] = X2(i, NXm);
}

// Builds the device constants.  Row placement: see mpc_kernel.h.
inline int buildMpcConst(const scpp_mpc_opts &o, const double *par, MpcConst &C)
{
    std::memset(&C, 0, sizeof C);
    const int K = o.K, N = K - 1;
    if (K < 3 || K > KMAX)
        return SCPP_E_ARG;
    // constant_dynamics = false only turns cvx::par(A) into cvx::dynpar(A) (MPCProblem.cpp:42-54); MPCAlgorithm never changes A, B, z
    // after initialize() (MPCAlgorithm.cpp:47), so both settings pose the same problem and are accepted
    if (o.nondimensionalize || o.intermediate_cost_active)
        return SCPP_E_UNSUPPORTED;
    if (!(o.time_horizon > 0.) || !(o.T_max > o.T_min) || !(o.gimbal_max > 0.) || !(o.theta_max > 0.) || !(o.w_B_max > 0.))
        return SCPP_E_ARG;
    C.K = K;
    C.N = N;
    C.nv = NU * N + 2;
    C.nlp = 8 * N;
    C.maxit = o.maxit > 0 ? o.maxit : 50;
    C.feastol = o.feastol > 0. ? o.feastol : 1e-8;
    C.abstol = o.abstol > 0. ? o.abstol : 1e-8;
    C.reltol = o.reltol > 0. ? o.reltol : 1e-8;
    C.tan_gs = o.tan_gamma_gs;
    C.theta_max = o.theta_max;
    C.w_max = o.w_B_max;
    exactLinearDiscretization<Rocket2dModel>(par, o.time_horizon / double(K - 1), o.x_eq, o.u_eq, C.A, C.B, C.z);
    // prediction matrices x_k = Phi_k x0 + sum_j Gam_kj u_j + zeta_k
    for (int i = 0; i < NX; i++)
        C.Phi[0][i][i] = 1.;
    for (int k = 1; k < K; k++)
    {
        for (int i = 0; i < NX; i++)
        {
            for (int j = 0; j < NX; j++)
                for (int l = 0; l < NX; l++)
                    C.Phi[k][i][j] += C.A[i * NX + l] * C.Phi[k - 1][l][j];
            C.zeta[k][i] = C.z[i];
            for (int l = 0; l < NX; l++)
                C.zeta[k][i] += C.A[i * NX + l] * C.zeta[k - 1][l];
            for (int j = 0; j + 1 < k; j++)
                for (int c = 0; c < NU; c++)
                    for (int l = 0; l < NX; l++)
                        C.Gam[k][j][i][c] += C.A[i * NX + l] * C.Gam[k - 1][j][l][c];
            for (int c = 0; c < NU; c++)
                C.Gam[k][k - 1][i][c] = C.B[i * NU + c];
        }
    }
    // unscaled rows  s = c0 + a' x_k + (direct variable terms) + af' x_final
    auto stateRow = [&](int row, int k, const double *ax, double cst, const double *af) {
        C.c0[row] = cst;
        for (int i = 0; i < NX; i++)
        {
            if (ax[i] == 0.)
                continue;
            C.c0[row] += ax[i] * C.zeta[k][i];
            for (int j = 0; j < NX; j++)
                C.P[row][j] += ax[i] * C.Phi[k][i][j];
            for (int j = 0; j < N; j++)
                for (int c = 0; c < NU; c++)
                    C.G[row][j * NU + c] -= ax[i] * C.Gam[k][j][i][c];
        }
        if (af)
            for (int i = 0; i < NX; i++)
                C.Q[row][i] = af[i];
    };
    auto varRow = [&](int row, int var, double coef, double cst) {
        C.c0[row] = cst;
        C.G[row][var] = -coef;
    };
    const int v_ic = NU * N, v_ec = NU * N + 1;
    int row = 0;
    for (int k = 1; k < K; k++)
    {
        const double tp[NX] = {0, 0, 0, 0, 1., 0}, tm[NX] = {0, 0, 0, 0, -1., 0};
        const double wp[NX] = {0, 0, 0, 0, 0, 1.}, wm[NX] = {0, 0, 0, 0, 0, -1.};
        stateRow(row++, k, tp, o.theta_max, nullptr);
        stateRow(row++, k, tm, o.theta_max, nullptr);
        stateRow(row++, k, wp, o.w_B_max, nullptr);
        stateRow(row++, k, wm, o.w_B_max, nullptr);
    }
    for (int j = 0; j < N; j++)
    {
        varRow(row++, j * NU + 0, 1., o.gimbal_max);
        varRow(row++, j * NU + 0, -1., o.gimbal_max);
        varRow(row++, j * NU + 1, 1., -o.T_min);
        varRow(row++, j * NU + 1, -1., o.T_max);
    }
    for (int k = 1; k < K; k++)
    {
        const double a1[NX] = {0, o.tan_gamma_gs, 0, 0, 0, 0}, a0[NX] = {1., 0, 0, 0, 0, 0};
        stateRow(64 + 2 * (k - 1), k, a1, 0., nullptr);
        stateRow(64 + 2 * (k - 1) + 1, k, a0, 0., nullptr);
    }
    varRow(64 + ERR_LANE, v_ec, 1., 0.);
    for (int i = 0; i < NX; i++)
    {
        double ax[NX] = {0, 0, 0, 0, 0, 0}, af[NX] = {0, 0, 0, 0, 0, 0};
        ax[i] = o.state_weights_terminal[i];
        af[i] = -o.state_weights_terminal[i];
        stateRow(64 + ERR_LANE + 1 + i, K - 1, ax, 0., af);
    }
    varRow(64 + INP_LANE, v_ic, 1., 0.);
    for (int j = 0; j < N; j++)
        for (int c = 0; c < NU; c++)
            varRow(64 + INP_LANE + 1 + j * NU + c, j * NU + c, o.input_weights[c], 0.);
    // ---- scaling: columns by physical magnitudes, rows (uniform per cone) to unit max-norm ----
    for (int j = 0; j < NV; j++)
        C.D[j] = 1.;
    double wtmax = 0.;
    for (int i = 0; i < NX; i++)
        wtmax = std::fmax(wtmax, std::fabs(o.state_weights_terminal[i]));
    for (int j = 0; j < N; j++)
    {
        C.D[j * NU + 0] = o.gimbal_max;
        C.D[j * NU + 1] = o.T_max;
    }
    C.D[v_ic] = std::fabs(o.input_weights[1]) * o.T_max;
    C.D[v_ec] = wtmax * o.x_scale_ref;
    if (!(C.D[v_ic] > 0.) || !(C.D[v_ec] > 0.))
        return SCPP_E_ARG;
    auto rowMax = [&](int r) {
        double mx = 0.;
        for (int v = 0; v < C.nv; v++)
            mx = std::fmax(mx, std::fabs(C.G[r][v] * C.D[v]));
        return mx;
    };
    auto scaleRows = [&](int first, int count) {
        double mx = 0.;
        for (int i = 0; i < count; i++)
            mx = std::fmax(mx, rowMax(first + i));
        const double e = 1. / mx;
        for (int i = 0; i < count; i++)
        {
            const int r = first + i;
            for (int v = 0; v < C.nv; v++)
                C.G[r][v] = e * C.G[r][v] * C.D[v];
            for (int q = 0; q < NX; q++)
            {
                C.P[r][q] *= e;
                C.Q[r][q] *= e;
            }
            C.c0[r] *= e;
        }
    };
    for (int r = 0; r < C.nlp; r++)
        scaleRows(r, 1);
    for (int k = 1; k < K; k++)
        scaleRows(64 + 2 * (k - 1), 2);
    scaleRows(64 + ERR_LANE, 1 + NX);
    scaleRows(64 + INP_LANE, 1 + NU * N);
    const double cs = std::fmax(C.D[v_ic], C.D[v_ec]);
    C.c[v_ic] = C.D[v_ic] / cs;
    C.c[v_ec] = C.D[v_ec] / cs;
    // ---- inverse Cholesky factor of H0 = G'G (identity on the padded variables) for the cold start ----
    double H[NV][NV] = {}, L[NV][NV] = {}, Li[NV][NV] = {};
    for (int r = 0; r < ROWS; r++)
        for (int a = 0; a < NV; a++)
        {
            if (C.G[r][a] == 0.)
                continue;
            for (int b2 = 0; b2 < NV; b2++)
                H[a][b2] += C.G[r][a] * C.G[r][b2];
        }
    for (int j = C.nv; j < NV; j++)
        H[j][j] = 1.;
    for (int j = 0; j < NV; j++)
    {
        double d = H[j][j];
        for (int k = 0; k < j; k++)
            d -= L[j][k] * L[j][k];
        if (!(d > 0.))
            return SCPP_E_ARG;
        L[j][j] = std::sqrt(d);
        for (int i = j + 1; i < NV; i++)
        {
            double v = H[i][j];
            for (int k = 0; k < j; k++)
                v -= L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
    for (int c = 0; c < NV; c++) // columns of L^-1 by forward substitution
        for (int i = c; i < NV; i++)
        {
            double v = i == c ? 1. : 0.;
            for (int k = c; k < i; k++)
                v -= L[i][k] * Li[k][c];
            Li[i][c] = v / L[i][i];
        }
    for (int i = 0; i < NV; i++)
        for (int j = 0; j < NV; j++)
        {
            C.Li0[i * NV + j] = Li[i][j];
            C.Li0T[j * NV + i] = Li[i][j];
        }
    return SCPP_OK;
}

} // namespace mpc
} // namespace scpp
