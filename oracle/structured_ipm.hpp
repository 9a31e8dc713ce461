// ORACLE (test infrastructure, NOT product code).
//
// Scalar CPU twin of the structure-exploiting interior-point method the HIP kernel
// (scpp_amd/csrc/ipm_kernel.hip) implements for the RocketQuat SC sub-problem
//   buildSCProblem                 scpp_core/src/SCProblem.cpp:6-138
//   RocketQuat::addApplicationConstraints  scpp_models/src/rocketQuat.cpp:70-144
// It solves the SAME optimisation problem as oracle/socp.hpp does on the literal standard form,
// but (i) presolves the fixed variables (x_0 = x_init, final-state components, U[{0,1},K-1]=0,
// roll off: X[13,:]=U[3,:]=0), (ii) eliminates nu, nu_bound, norm1_nu, delta_k, delta_sigma
// analytically and (iii) factorises the remaining block-tridiagonal quasi-definite KKT system
// (16x16 stage blocks, 14x14 multiplier blocks, sigma as a border) with dense Cholesky.
// Algorithm: primal-dual Mehrotra predictor-corrector with Nesterov-Todd scaling (as ECOS),
// WITHOUT the self-dual embedding (sub-problems are feasible by construction: virtual control).
// Used (a) as the numerically robust CPU oracle for full SC solves and (b) as the line-by-line
// parity reference of the HIP kernel.
// Parity status: UNPINNED at the ECOS boundary (see socp.hpp); cross-checked against the literal standard-form solver
// (first sub-problem 2e-6, whole SC run at K = 15) and used as the iterate-level twin of the HIP kernel.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace oracle
{

struct RQStageData // constants of the RocketQuat stage constraints (nondimensional)
{
    double gs, tilt, wmax, Tmin, Tmax, gim, mdry;
};

struct RQSocpInput
{
    int K;
    const double *Xbar;                 // [K][14] trust-region centre / linearisation point
    const double *Ubar;                 // [K][4]
    double sigbar;                      // sigma_0
    const double *A, *B, *C, *S, *Z;    // dd, row-major per segment
    const double *x_init, *x_final;     // [14]
    const double *uhat;                 // [K][3] linearised min-thrust directions (thrust_const)
    RQStageData cst;
    double w_t, w_trt, w_trx, w_vc;     // cost weights
    // SCvx mode (SCvxProblem.cpp:6-71): hard INPUT trust region ||u_k - ubar_k|| <= trust_region instead of the
    // penalised state+input one, fixed final time.  Realised inside the same structure: delta_k is a constant
    // (= trust_region, no elimination), the state rows of the trust cone are zero padding (same central path as the
    // 4-dimensional cone: the barrier does not see zero components), S = 0 decouples the (then dummy) sigma block.
    bool scvx = false;
    double trust_region = 0.;
};

struct RQSocpSettings
{
    double feastol = 1e-8, abstol = 1e-7, reltol = 1e-7;
    int maxit = 60;
    double gamma = 0.99;
    // step lengths of their own for the primal variables (x, s) and the dual ones (multipliers, z): the device kernel's IPM_SPLIT_STEPS (round 6;
    // csrc/ipm_solve.h).  false: ECOS's single step length, the largest that keeps s AND z in their cones (rounds 1 - 6a).  The centring
    // parameter follows ECOS's rule on the COMMON affine step length either way.  tools/experiments/split_step_study.py: 366 -> 327
    // interior-point iterations per RocketQuat K = 50 SCvx trajectory.
    bool split_steps = true;
    double rfloor = 0.;       // optional floor on s/z of the virtual-control rows (0 = exact Newton) // floor on the nu Hessian inverse (inexact-Newton safeguard)
    bool verbose = false;
};

struct RQSocpOutput
{
    std::vector<double> X, U, nu;
    double sigma, delta_sigma, norm1_nu, sum_delta;
    std::vector<double> delta;
    int iters, status; // 0 ok, -1 maxit, -2 numerics
    double pres, dres, gap, pcost;
};

namespace sipm
{
constexpr int NX = 14, NU = 4, NV = 16, NS = 35, NL = 14;
// stage cone table: offset, dim
constexpr int C1 = 0, C2 = 17, C3 = 20, C4 = 23, C5 = 26, C6 = 30, L1 = 33, L2 = 34;
constexpr int NCONE = 6;
constexpr int cone_off[NCONE] = {C1, C2, C3, C4, C5, C6};
constexpr int cone_dim[NCONE] = {17, 3, 3, 3, 4, 3};

inline double sq(double a) { return a * a; }

// --- second-order-cone helpers on small arrays (dim d) ---
struct Scaling
{
    double eta, w[17];
};
inline bool nt_scaling(const double *s, const double *z, int d, Scaling &sc)
{
    double s1 = 0., z1 = 0.;
    for (int i = 1; i < d; i++)
    {
        s1 += s[i] * s[i];
        z1 += z[i] * z[i];
    }
    const double sres = s[0] * s[0] - s1, zres = z[0] * z[0] - z1;
    if (!(sres > 0.) || !(zres > 0.))
        return false;
    const double sn = std::sqrt(sres), zn = std::sqrt(zres);
    double sz = 0.;
    for (int i = 0; i < d; i++)
        sz += (s[i] / sn) * (z[i] / zn);
    const double gamma = std::sqrt(0.5 * (1. + sz));
    const double a = 0.5 / gamma;
    sc.w[0] = a * (s[0] / sn + z[0] / zn);
    for (int i = 1; i < d; i++)
        sc.w[i] = a * (s[i] / sn - z[i] / zn);
    sc.eta = std::sqrt(sn / zn);
    return true;
}
inline void applyW(const Scaling &sc, int d, const double *v, double *out)
{
    double zeta = 0.;
    for (int i = 1; i < d; i++)
        zeta += sc.w[i] * v[i];
    const double f = v[0] + zeta / (1. + sc.w[0]);
    out[0] = sc.eta * (sc.w[0] * v[0] + zeta);
    for (int i = 1; i < d; i++)
        out[i] = sc.eta * (v[i] + f * sc.w[i]);
}
inline void applyWinv(const Scaling &sc, int d, const double *v, double *out)
{
    double zeta = 0.;
    for (int i = 1; i < d; i++)
        zeta += sc.w[i] * v[i];
    const double f = -v[0] + zeta / (1. + sc.w[0]);
    out[0] = (sc.w[0] * v[0] - zeta) / sc.eta;
    for (int i = 1; i < d; i++)
        out[i] = (v[i] + f * sc.w[i]) / sc.eta;
}
// W^-2 v = (2 vt (vt'v) - J v)/eta^2, vt = (w0, -w1)
inline void applyWinv2(const Scaling &sc, int d, const double *v, double *out)
{
    double tv = sc.w[0] * v[0];
    for (int i = 1; i < d; i++)
        tv -= sc.w[i] * v[i];
    const double e2 = 1. / (sc.eta * sc.eta);
    out[0] = e2 * (2. * sc.w[0] * tv - v[0]);
    for (int i = 1; i < d; i++)
        out[i] = e2 * (-2. * sc.w[i] * tv + v[i]);
}
inline void conicProduct(int d, const double *u, const double *v, double *out)
{
    double s0 = 0.;
    for (int i = 0; i < d; i++)
        s0 += u[i] * v[i];
    const double u0 = u[0], v0 = v[0];
    for (int i = 1; i < d; i++)
        out[i] = u0 * v[i] + v0 * u[i];
    out[0] = s0;
}
inline void conicDivision(int d, const double *lam, const double *dd, double *out)
{
    double l1d1 = 0., l1l1 = 0.;
    for (int i = 1; i < d; i++)
    {
        l1d1 += lam[i] * dd[i];
        l1l1 += lam[i] * lam[i];
    }
    const double rho = lam[0] * lam[0] - l1l1;
    const double u0 = (lam[0] * dd[0] - l1d1) / rho;
    for (int i = 1; i < d; i++)
        out[i] = (dd[i] - u0 * lam[i]) / lam[0];
    out[0] = u0;
}
// 1/alpha_max for lambda + alpha*v staying in the cone
inline double stepInv(int d, const double *lam, const double *v)
{
    double l1 = 0.;
    for (int i = 1; i < d; i++)
        l1 += lam[i] * lam[i];
    const double ln = std::sqrt(lam[0] * lam[0] - l1);
    double lbJv = lam[0] * v[0];
    for (int i = 1; i < d; i++)
        lbJv -= lam[i] * v[i];
    lbJv /= ln;
    const double rho0 = lbJv / ln;
    const double f = (lbJv + v[0]) / (lam[0] / ln + 1.);
    double r1 = 0.;
    for (int i = 1; i < d; i++)
    {
        const double ri = (v[i] - f * lam[i] / ln) / ln;
        r1 += ri * ri;
    }
    return std::sqrt(r1) - rho0;
}

// Inverse Cholesky factor of an SPD n x n matrix by Gaussian elimination on the augmented array [A | I]
// (no pivoting): A = Lt D Lt', the right half becomes Lt^-1 (unit lower), and Li = D^-1/2 Lt^-1 = L^-1 with
// A = L L'.  Every elimination step is a rank-1 update of BOTH halves, so the HIP tile engine runs it with all
// 64 lanes busy; forward error ~ sqrt(cond(A)) eps (triangular factor), unlike an explicit A^-1.
// Pivot floor 1e-14 * original diagonal (multiplier block of segment 0 is tiny-diagonal + rank 3 near convergence).
// Li: row-major n x n (ld), lower triangular.  Returns false on non-finite input.
inline bool invCholFactor(int n, const double *Ain, int ld, double *Li)
{
    double A[16 * 16], R[16 * 16], orig[16];
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++)
        {
            A[r * 16 + c] = Ain[r * ld + c];
            R[r * 16 + c] = (r == c) ? 1. : 0.;
        }
    for (int j = 0; j < n; j++)
    {
        orig[j] = A[j * 16 + j];
        if (!(orig[j] > 0.) || !std::isfinite(orig[j]))
            return false;
    }
    double piv[16];
    for (int j = 0; j < n; j++)
    {
        double d = A[j * 16 + j];
        if (!std::isfinite(d))
            return false;
        if (!(d > 1e-14 * orig[j]))
            d = 1e-14 * orig[j];
        piv[j] = d;
        const double p = 1. / d;
        for (int r = j + 1; r < n; r++)
        {
            const double m = A[r * 16 + j] * p; // multiplier (column j of A == row j by symmetry of the trailing part)
            for (int c = j + 1; c < n; c++)
                A[r * 16 + c] -= m * A[j * 16 + c];
            for (int c = 0; c <= j; c++)
                R[r * 16 + c] -= m * R[j * 16 + c];
        }
    }
    for (int r = 0; r < n; r++)
    {
        const double sc = 1. / std::sqrt(piv[r]);
        for (int c = 0; c < n; c++)
            Li[r * ld + c] = (c <= r) ? R[r * 16 + c] * sc : 0.;
    }
    return true;
}
} // namespace sipm

class RQStructuredSocp
{
  public:
    RQSocpSettings opt;
    int last_fail = 0;

    // warm = true: start the interior-point iteration from the previous solve's primal-dual point (kept in this
    // object) shifted back into the cone interior, instead of ECOS's cold initialisation
    RQSocpOutput solve(const RQSocpInput &in, bool warm = false)
    {
        using namespace sipm;
        P = &in;
        warm_start = warm && have_prev && K == in.K;
        restored_best = false;
        K = in.K;
        if (!warm_start)
            alloc();
        setupStages();
        RQSocpOutput out;
        // a warm-started solve of SCAlgorithm's sub-problem takes ECOS's common step length from the start (csrc/ipm_solve.h: ipmSolveInstance; measured on the
        // GPU in SC_sim's shape); cold solves and every SCvx solve take primal and dual step lengths of their own
        const bool split_cfg = opt.split_steps;
        if (warm_start && !in.scvx)
            opt.split_steps = false;
        out.status = run(out);
        if (out.status != 0 && warm_start)
        {
            // a warm start that breaks down is repeated from ECOS's cold initialisation (with the cold attempt's own step-length rule)
            opt.split_steps = split_cfg;
            warm_start = false;
            restored_best = false;
            alloc();
            setupStages();
            const int warm_iters = out.iters;
            out = RQSocpOutput();
            out.status = run(out);
            out.iters += warm_iters;
        }
        if (out.status == 0 || !opt.split_steps)
            opt.split_steps = split_cfg;
        else
        {
            // a cold attempt that fails with primal and dual step lengths of their own is repeated with ECOS's common step length
            // (csrc/ipm_solve.h: ipmSolveInstance, round 6: one of 1 048 576 soak trajectories on the device)
            opt.split_steps = false;
            warm_start = false;
            restored_best = false;
            alloc();
            setupStages();
            const int failed_iters = out.iters;
            out = RQSocpOutput();
            out.status = run(out);
            out.iters += failed_iters;
            opt.split_steps = split_cfg;
        }
        out.X.assign(size_t(K) * NX, 0.);
        out.U.assign(size_t(K) * NU, 0.);
        out.nu = nu;
        out.delta = dl;
        for (int k = 0; k < K; k++)
        {
            double x[NX], u[NU];
            unpack(k, &w[size_t(k) * NV], x, u);
            for (int i = 0; i < NX; i++)
                out.X[size_t(k) * NX + i] = x[i];
            for (int i = 0; i < NU; i++)
                out.U[size_t(k) * NU + i] = u[i];
        }
        out.sigma = sig;
        out.delta_sigma = dsg;
        out.norm1_nu = n1;
        out.sum_delta = 0.;
        for (int k = 0; k < K; k++)
            out.sum_delta += dl[k];
        have_prev = out.status == 0 && !restored_best;
        return out;
    }
    bool have_prev = false, warm_start = false, restored_best = false;
    double warm_theta = 1e-2;

  private:
    const RQSocpInput *P = nullptr;
    int K = 0;
    // primal
    std::vector<double> w, dl, nu, nub;
    double sig = 0, dsg = 0, n1 = 0;
    // cone slacks / duals
    std::vector<double> s, z;          // [K][NS]
    std::vector<double> s1, z1, s2, z2; // [K-1][14]
    double ss = 0, zs = 0;             // sigma >= 0.001
    double s3 = 0, z3 = 0;             // n1 - sum(nub) >= 0
    double sc3[3], zc3[3];             // sigma trust region cone
    std::vector<double> lam;           // [K-1][14]
    // stage meta
    std::vector<unsigned> fm;          // fixed mask over the 16 stage variables
    std::vector<unsigned> act;         // active cone mask: bits 0..5 cones C1..C6, bit 6 L1, bit 7 L2
    std::vector<double> wbar;          // [K][16] trust region centre in stage coords
    // scalings
    std::vector<sipm::Scaling> scal;   // [K][6]
    sipm::Scaling scsig;
    // directions
    std::vector<double> dw, ddl, dnu, dnub, dlam, ds, dz, ds1, dz1, ds2, dz2;
    double dsig = 0, ddsg = 0, dn1 = 0, dss = 0, dzs = 0, ds3 = 0, dz3 = 0, dsc3[3], dzc3[3];
    // factor storage
    std::vector<double> Lif, Ytf, Tif, Zf; // [K][256]: Li = chol(Phi)^-1 (16x16), Yt = Li M' (16x14, ld 16), Ti = chol(Theta)^-1 (14x14, ld 14), Z = Ti N (14x16)
    std::vector<double> Einv;            // [K-1][14]
    std::vector<double> qv;              // [K-1][14]  (d2-d1)/(d1+d2)
    std::vector<double> hdd, hdw;        // [K], [K][16]: delta_k elimination
    double hsig = 0, Hsd = 0, Hdd = 0;
    std::vector<double> bcol_w, bcol_l;  // T^-1 c_sigma
    double schur_sig = 0;
    int D = 0;

    void alloc()
    {
        using namespace sipm;
        w.assign(size_t(K) * NV, 0.);
        dl.assign(K, 0.);
        nu.assign(size_t(K - 1) * NL, 0.);
        nub.assign(size_t(K - 1) * NL, 0.);
        s.assign(size_t(K) * NS, 0.);
        z.assign(size_t(K) * NS, 0.);
        s1.assign(size_t(K - 1) * NL, 0.);
        z1 = s2 = z2 = s1;
        lam.assign(size_t(K - 1) * NL, 0.);
        fm.assign(K, 0u);
        act.assign(K, 0xFFu);
        wbar.assign(size_t(K) * NV, 0.);
        scal.resize(size_t(K) * NCONE);
        dw = w;
        ddl = dl;
        dnu = nu;
        dnub = nub;
        dlam = lam;
        ds = s;
        dz = z;
        ds1 = dz1 = ds2 = dz2 = s1;
        Lif.assign(size_t(K) * 256, 0.);
        Ytf = Tif = Zf = Lif;
        Einv.assign(size_t(K - 1) * NL, 0.);
        qv = Einv;
        hdd.assign(K, 0.);
        hdw.assign(size_t(K) * NV, 0.);
        bcol_w.assign(size_t(K) * NV, 0.);
        bcol_l.assign(size_t(K - 1) * NL, 0.);
    }

    // stage variable j -> (x index) for j<13, (u index) for j>=13
    static void unpack(int, const double *wk, double *x, double *u)
    {
        for (int j = 0; j < 13; j++)
            x[j] = wk[j];
        x[13] = 0.;
        for (int j = 0; j < 3; j++)
            u[j] = wk[13 + j];
        u[3] = 0.;
    }

    void setupStages()
    {
        using namespace sipm;
        for (int k = 0; k < K; k++)
        {
            for (int j = 0; j < 13; j++)
                wbar[size_t(k) * NV + j] = P->Xbar[size_t(k) * NX + j];
            for (int j = 0; j < 3; j++)
                wbar[size_t(k) * NV + 13 + j] = P->Ubar[size_t(k) * NU + j];
        }
        // stage 0: x fixed
        fm[0] = 0x1FFFu;
        for (int j = 0; j < 13; j++)
            w[j] = P->x_init[j];
        act[0] = 0xFFu & ~((1u << 1) | (1u << 2) | (1u << 3) | (1u << 6));
        // stage K-1: final-state components, U[0,1] = 0
        unsigned m = 0;
        for (int i : {1, 2, 3, 4, 5, 6, 8, 9, 11, 12})
        {
            m |= 1u << i;
            w[size_t(K - 1) * NV + i] = P->x_final[i];
        }
        m |= (1u << 13) | (1u << 14);
        fm[K - 1] |= m;
        act[K - 1] &= ~((1u << 1) | (1u << 2) | (1u << 3));
        D = 0;
        for (int k = 0; k < K; k++)
            for (int b = 0; b < 8; b++)
                if (act[k] & (1u << b))
                    D++;
        D += 2 * NL * (K - 1) + 1 /*sigma lp*/ + 1 /*r3*/ + 1 /*sigma cone*/;
    }

    // affine slack s_aff(x) = h - Gx of stage k
    void saff(int k, const double *wk, double dlk, double *out) const
    {
        using namespace sipm;
        const double *wb = &wbar[size_t(k) * NV];
        const double *uh = &P->uhat[size_t(k) * 3];
        const RQStageData &c = P->cst;
        out[0] = P->scvx ? P->trust_region : dlk;
        for (int j = 0; j < NV; j++)
            out[1 + j] = tx(j) * (wb[j] - wk[j]);
        out[17] = c.gs * wk[3];
        out[18] = wk[1];
        out[19] = wk[2];
        out[20] = c.tilt;
        out[21] = wk[8];
        out[22] = wk[9];
        out[23] = c.wmax;
        out[24] = wk[11];
        out[25] = wk[12];
        out[26] = c.Tmax;
        out[27] = wk[13];
        out[28] = wk[14];
        out[29] = wk[15];
        out[30] = c.gim * wk[15];
        out[31] = wk[13];
        out[32] = wk[14];
        out[33] = wk[0] - c.mdry;
        out[34] = uh[0] * wk[13] + uh[1] * wk[14] + uh[2] * wk[15] - c.Tmin;
        maskInactive(k, out);
    }
    // static dual regularisation -delta I of the multiplier block (every interior-point code has one; ECOS: 7e-8 with
    // iterative refinement).  SCvx only: there the virtual control really goes to zero (E^-1 -> 0), and with the
    // initial state fixed M_0 has rank 3 < 14, so Theta_0 = E^-1 + Y Y' would become numerically singular.
    double dualReg() const
    {
        if (!P->scvx)
            return 0.;
        const char *e = std::getenv("ORACLE_SCVX_DUALREG");
        return e ? std::atof(e) : 1e-9;
    }
    // trust-cone row of stage variable j present?  (SCvx: inputs only)
    double tx(int j) const { return (P->scvx && j < 13) ? 0. : 1.; }
    void maskInactive(int k, double *v) const
    {
        using namespace sipm;
        for (int cix = 0; cix < NCONE; cix++)
            if (!(act[k] & (1u << cix)))
                for (int i = 0; i < cone_dim[cix]; i++)
                    v[cone_off[cix] + i] = 0.;
        if (!(act[k] & (1u << 6)))
            v[L1] = 0.;
        if (!(act[k] & (1u << 7)))
            v[L2] = 0.;
    }
    // L(dx): linear part of saff
    void Lmul(int k, const double *dwk, double ddlk, double *out) const
    {
        using namespace sipm;
        const double *uh = &P->uhat[size_t(k) * 3];
        const RQStageData &c = P->cst;
        out[0] = P->scvx ? 0. : ddlk;
        for (int j = 0; j < NV; j++)
            out[1 + j] = -tx(j) * dwk[j];
        out[17] = c.gs * dwk[3];
        out[18] = dwk[1];
        out[19] = dwk[2];
        out[20] = 0.;
        out[21] = dwk[8];
        out[22] = dwk[9];
        out[23] = 0.;
        out[24] = dwk[11];
        out[25] = dwk[12];
        out[26] = 0.;
        out[27] = dwk[13];
        out[28] = dwk[14];
        out[29] = dwk[15];
        out[30] = c.gim * dwk[15];
        out[31] = dwk[13];
        out[32] = dwk[14];
        out[33] = dwk[0];
        out[34] = uh[0] * dwk[13] + uh[1] * dwk[14] + uh[2] * dwk[15];
        maskInactive(k, out);
    }
    // L' v  -> gw[16], gdl   (v of inactive cones must be zero)
    void LTmul(int k, const double *v, double *gw, double &gdl) const
    {
        using namespace sipm;
        const double *uh = &P->uhat[size_t(k) * 3];
        const RQStageData &c = P->cst;
        gdl = v[0];
        for (int j = 0; j < NV; j++)
            gw[j] = -tx(j) * v[1 + j];
        gw[3] += c.gs * v[17];
        gw[1] += v[18];
        gw[2] += v[19];
        gw[8] += v[21];
        gw[9] += v[22];
        gw[11] += v[24];
        gw[12] += v[25];
        gw[13] += v[27] + v[31] + uh[0] * v[34];
        gw[14] += v[28] + v[32] + uh[1] * v[34];
        gw[15] += v[29] + c.gim * v[30] + uh[2] * v[34];
        gw[0] += v[33];
        for (int j = 0; j < NV; j++)
            if (fm[k] & (1u << j))
                gw[j] = 0.;
    }

    // dynamics residual of segment k:  x_{k+1} - A x_k - B u_k - C u_{k+1} - S sigma - nu_k - Z_k
    void dynRes(int k, const double *wv, const double *nuv, double sg, double *out) const
    {
        using namespace sipm;
        double x0[NX], u0[NU], x1[NX], u1[NU];
        unpack(k, &wv[size_t(k) * NV], x0, u0);
        unpack(k + 1, &wv[size_t(k + 1) * NV], x1, u1);
        const double *A = &P->A[size_t(k) * NX * NX], *B = &P->B[size_t(k) * NX * NU], *C = &P->C[size_t(k) * NX * NU];
        for (int i = 0; i < NX; i++)
        {
            double acc = x1[i] - P->S[size_t(k) * NX + i] * sg - nuv[size_t(k) * NL + i] - P->Z[size_t(k) * NX + i];
            for (int j = 0; j < NX; j++)
                acc -= A[i * NX + j] * x0[j];
            for (int j = 0; j < NU; j++)
                acc -= B[i * NU + j] * u0[j] + C[i * NU + j] * u1[j];
            out[i] = acc;
        }
    }
    // M_k (14x16): coefficient of stage-k variables in segment k;  N_k: of stage-(k+1) variables
    void buildM(int k, double *M) const
    {
        using namespace sipm;
        const double *A = &P->A[size_t(k) * NX * NX], *B = &P->B[size_t(k) * NX * NU];
        for (int i = 0; i < NL; i++)
        {
            for (int j = 0; j < 13; j++)
                M[i * NV + j] = (fm[k] & (1u << j)) ? 0. : -A[i * NX + j];
            for (int j = 0; j < 3; j++)
                M[i * NV + 13 + j] = (fm[k] & (1u << (13 + j))) ? 0. : -B[i * NU + j];
        }
    }
    void buildN(int k, double *N) const
    {
        using namespace sipm;
        const double *C = &P->C[size_t(k) * NX * NU];
        for (int i = 0; i < NL; i++)
        {
            for (int j = 0; j < 13; j++)
                N[i * NV + j] = ((fm[k + 1] & (1u << j)) || i != j) ? 0. : 1.;
            for (int j = 0; j < 3; j++)
                N[i * NV + 13 + j] = (fm[k + 1] & (1u << (13 + j))) ? 0. : -C[i * NU + j];
        }
    }

    // ---- scalings + factorisation -------------------------------------------------------------
    bool identityScaling = false;
    bool updateScalings()
    {
        using namespace sipm;
        for (int k = 0; k < K; k++)
            for (int c = 0; c < NCONE; c++)
                if (act[k] & (1u << c))
                    if (!nt_scaling(&s[size_t(k) * NS + cone_off[c]], &z[size_t(k) * NS + cone_off[c]], cone_dim[c],
                                    scal[size_t(k) * NCONE + c]))
                    {
                        if (opt.verbose)
                        {
                            std::printf("nt_scaling failed: stage %d cone %d\n  s:", k, c);
                            for (int i = 0; i < cone_dim[c]; i++)
                                std::printf(" %.3e", s[size_t(k) * NS + cone_off[c] + i]);
                            std::printf("\n  z:");
                            for (int i = 0; i < cone_dim[c]; i++)
                                std::printf(" %.3e", z[size_t(k) * NS + cone_off[c] + i]);
                            std::printf("\n");
                        }
                        return false;
                    }
        if (!nt_scaling(sc3, zc3, 3, scsig))
            return false;
        return true;
    }
    void setIdentityScalings()
    {
        using namespace sipm;
        for (auto &sc : scal)
        {
            sc.eta = 1.;
            for (int i = 0; i < 17; i++)
                sc.w[i] = 0.;
            sc.w[0] = 1.;
        }
        scsig.eta = 1.;
        scsig.w[0] = 1.;
        scsig.w[1] = scsig.w[2] = 0.;
    }
    // LP "d" = z/s (or 1 for the W=I initialisation solves)
    double dLP(double sv, double zv) const { return identityScaling ? 1. : zv / sv; }
    // virtual-control rows: d = z/s capped at 1/rfloor, used CONSISTENTLY in H, t and dz
    double dNu(double sv, double zv) const { return identityScaling ? 1. : 1. / std::max(sv / zv, opt.rfloor); }

    // H += sum_ab c_a c_b W^-2_ab e_va e_vb'
    static void addConeH(double *H, const sipm::Scaling &sc, int d, const int *vars, const double *coef)
    {
        using namespace sipm;
        const double e2 = 1. / (sc.eta * sc.eta);
        for (int a = 0; a < d; a++)
        {
            if (vars[a] < 0)
                continue;
            const double va = (a == 0) ? sc.w[0] : -sc.w[a];
            for (int b = 0; b < d; b++)
            {
                if (vars[b] < 0)
                    continue;
                const double vb = (b == 0) ? sc.w[0] : -sc.w[b];
                double Wab = 2. * va * vb;
                if (a == b)
                    Wab += (a == 0) ? -1. : 1.;
                H[vars[a] * NV + vars[b]] += coef[a] * coef[b] * Wab * e2;
            }
        }
    }

    void buildH(int k, double *H)
    {
        using namespace sipm;
        const RQStageData &c = P->cst;
        for (int i = 0; i < NV * NV; i++)
            H[i] = 0.;
        // C1 with delta_k eliminated
        {
            const Scaling &sc = scal[size_t(k) * NCONE + 0];
            const double e2 = 1. / (sc.eta * sc.eta);
            const double den = 2. * sc.w[0] * sc.w[0] - 1.;
            if (!P->scvx)
            {
                hdd[k] = den * e2;
                for (int j = 0; j < NV; j++)
                    hdw[size_t(k) * NV + j] = (fm[k] & (1u << j)) ? 0. : 2. * sc.w[0] * sc.w[1 + j] * e2;
                for (int i = 0; i < NV; i++)
                    for (int j = 0; j < NV; j++)
                        H[i * NV + j] = e2 * ((i == j ? 1. : 0.) - (2. / den) * sc.w[1 + i] * sc.w[1 + j]);
            }
            else
            {
                // delta is a constant: plain L' W^-2 L of the rows that exist, W^-2_ab = e2 (delta_ab + 2 w_a w_b), a,b >= 1
                hdd[k] = 1.;
                for (int j = 0; j < NV; j++)
                    hdw[size_t(k) * NV + j] = 0.;
                for (int i = 0; i < NV; i++)
                    for (int j = 0; j < NV; j++)
                        H[i * NV + j] = tx(i) * tx(j) * e2 * ((i == j ? 1. : 0.) + 2. * sc.w[1 + i] * sc.w[1 + j]);
            }
        }
        if (act[k] & 2u)
        {
            const int v[3] = {3, 1, 2};
            const double cf[3] = {c.gs, 1., 1.};
            addConeH(H, scal[size_t(k) * NCONE + 1], 3, v, cf);
        }
        if (act[k] & 4u)
        {
            const int v[3] = {-1, 8, 9};
            const double cf[3] = {0., 1., 1.};
            addConeH(H, scal[size_t(k) * NCONE + 2], 3, v, cf);
        }
        if (act[k] & 8u)
        {
            const int v[3] = {-1, 11, 12};
            const double cf[3] = {0., 1., 1.};
            addConeH(H, scal[size_t(k) * NCONE + 3], 3, v, cf);
        }
        if (act[k] & 16u)
        {
            const int v[4] = {-1, 13, 14, 15};
            const double cf[4] = {0., 1., 1., 1.};
            addConeH(H, scal[size_t(k) * NCONE + 4], 4, v, cf);
        }
        if (act[k] & 32u)
        {
            const int v[3] = {15, 13, 14};
            const double cf[3] = {c.gim, 1., 1.};
            addConeH(H, scal[size_t(k) * NCONE + 5], 3, v, cf);
        }
        if (act[k] & 64u)
            H[0] += dLP(s[size_t(k) * NS + L1], z[size_t(k) * NS + L1]);
        if (act[k] & 128u)
        {
            const double d = dLP(s[size_t(k) * NS + L2], z[size_t(k) * NS + L2]);
            const double *uh = &P->uhat[size_t(k) * 3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++)
                    H[(13 + a) * NV + 13 + b] += d * uh[a] * uh[b];
        }
        // fixed variables: identity rows
        for (int j = 0; j < NV; j++)
            if (fm[k] & (1u << j))
            {
                for (int i = 0; i < NV; i++)
                    H[i * NV + j] = H[j * NV + i] = 0.;
                H[j * NV + j] = 1.;
            }
    }

    bool factor()
    {
        using namespace sipm;
        // segment scalars
        for (int k = 0; k < K - 1; k++)
            for (int i = 0; i < NL; i++)
            {
                const size_t o = size_t(k) * NL + i;
                if (identityScaling)
                {
                    Einv[o] = 0.5; // d1=d2=1: e = 4*1*1/2 = 2
                    qv[o] = 0.;
                }
                else
                {
                    const double r1 = 1. / dNu(s1[o], z1[o]), r2 = 1. / dNu(s2[o], z2[o]); // 1/d1, 1/d2
                    Einv[o] = 0.25 * (r1 + r2);
                    // q = (d2-d1)/(d1+d2) = (r1-r2)/(r1+r2)
                    qv[o] = (r1 - r2) / (r1 + r2);
                }
            }
        // sigma block
        {
            const double e2 = 1. / (scsig.eta * scsig.eta);
            const double vt[3] = {scsig.w[0], -scsig.w[1], -scsig.w[2]};
            auto Wi2 = [&](int a, int b) { return e2 * (2. * vt[a] * vt[b] + (a == b ? (a == 0 ? -1. : 1.) : 0.)); };
            // L rows (cone entries) wrt (sigma, dsg): [0,.5],[0,-.5],[1,0]
            const double Ls[3] = {0., 0., 1.}, Ld[3] = {0.5, -0.5, 0.};
            double Hss = 0., Hsd_ = 0., Hdd_ = 0.;
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++)
                {
                    Hss += Ls[a] * Wi2(a, b) * Ls[b];
                    Hsd_ += Ls[a] * Wi2(a, b) * Ld[b];
                    Hdd_ += Ld[a] * Wi2(a, b) * Ld[b];
                }
            Hss += dLP(ss, zs);
            Hsd = Hsd_;
            Hdd = Hdd_;
            hsig = Hss - Hsd * Hsd / Hdd;
        }
        // block-tridiagonal elimination; every stage operation is an X'Y product of 16x16 tiles (what the FP64
        // matrix cores of the GPU twin execute) except the two inverse-Cholesky factors:
        //   Phi_k = H_k + Z_{k-1}'Z_{k-1},  Li_k = chol(Phi_k)^-1,  Yt_k = Li_k M_k'   (= Y_k', Y_k = M_k L_k^-T)
        //   Theta_k = E_k^-1 + Yt_k'Yt_k,   Ti_k = chol(Theta_k)^-1, Z_k = Ti_k N_k
        double Phi[256], M[NL * NV], N[NL * NV], Th[256];
        for (int k = 0; k < K; k++)
        {
            buildH(k, Phi);
            if (k > 0)
            {
                const double *Zp = &Zf[size_t(k - 1) * 256];
                for (int i = 0; i < NV; i++)
                    for (int j = 0; j < NV; j++)
                    {
                        double acc = 0.;
                        for (int r = 0; r < NL; r++)
                            acc += Zp[r * NV + i] * Zp[r * NV + j];
                        Phi[i * NV + j] += acc;
                    }
            }
            double *Li = &Lif[size_t(k) * 256];
            if (!invCholFactor(NV, Phi, NV, Li))
            {
                last_fail = 100 + k;
                return false;
            }
            if (k == K - 1)
                break;
            buildM(k, M);
            // Yt = Li M'  (16 x 14), stored [var j][dyn row i]
            double *Yt = &Ytf[size_t(k) * 256];
            for (int j = 0; j < NV; j++)
                for (int i = 0; i < NL; i++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NV; q++)
                        acc += Li[j * NV + q] * M[i * NV + q];
                    Yt[j * NV + i] = acc;
                }
            for (int i = 0; i < NL; i++)
                for (int j = 0; j < NL; j++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NV; q++)
                        acc += Yt[q * NV + i] * Yt[q * NV + j];
                    if (i == j)
                        acc += Einv[size_t(k) * NL + i] + dualReg();
                    Th[i * NL + j] = acc;
                }
            double *Ti = &Tif[size_t(k) * 256];
            if (!invCholFactor(NL, Th, NL, Ti))
            {
                last_fail = 200 + k;
                return false;
            }
            buildN(k, N);
            // Z = Ti N  (14 x 16)
            double *Zk = &Zf[size_t(k) * 256];
            for (int i = 0; i < NL; i++)
                for (int j = 0; j < NV; j++)
                {
                    double acc = 0.;
                    for (int q = 0; q < NL; q++)
                        acc += Ti[i * NL + q] * N[q * NV + j];
                    Zk[i * NV + j] = acc;
                }
        }
        // border column: T_mat v = c_sigma, c_sigma = (0 in w rows, -S_k in lambda rows)
        {
            std::vector<double> bw(size_t(K) * NV, 0.), bl(size_t(K - 1) * NL);
            for (int k = 0; k < K - 1; k++)
                for (int i = 0; i < NL; i++)
                    bl[size_t(k) * NL + i] = -P->S[size_t(k) * NX + i];
            blockSolve(bw, bl, bcol_w, bcol_l);
            double acc = 0.;
            for (int k = 0; k < K - 1; k++)
                for (int i = 0; i < NL; i++)
                    acc += -P->S[size_t(k) * NX + i] * bcol_l[size_t(k) * NL + i];
            schur_sig = hsig - acc;
            if (!(schur_sig > 0.))
            {
                last_fail = 300;
                return false;
            }
        }
        return true;
    }

    // solve block-tridiagonal T_mat [dw; dlam] = [beta; rho] with the stored tiles Li, Yt, Ti, Z
    void blockSolve(const std::vector<double> &beta, const std::vector<double> &rho, std::vector<double> &ow,
                    std::vector<double> &ol) const
    {
        using namespace sipm;
        std::vector<double> av(size_t(K) * NV), cv(size_t(K - 1) * NL);
        double g[NV];
        for (int j = 0; j < NV; j++)
            g[j] = beta[j];
        for (int k = 0; k < K; k++)
        {
            const double *Li = &Lif[size_t(k) * 256];
            double *a = &av[size_t(k) * NV];
            for (int i = 0; i < NV; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NV; q++)
                    acc += Li[i * NV + q] * g[q];
                a[i] = acc;
            }
            if (k == K - 1)
                break;
            const double *Yt = &Ytf[size_t(k) * 256], *Ti = &Tif[size_t(k) * 256], *Zk = &Zf[size_t(k) * 256];
            double gl[NL];
            for (int i = 0; i < NL; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NV; q++)
                    acc += Yt[q * NV + i] * a[q];
                gl[i] = rho[size_t(k) * NL + i] - acc;
            }
            double *c = &cv[size_t(k) * NL];
            for (int i = 0; i < NL; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NL; q++)
                    acc += Ti[i * NL + q] * gl[q];
                c[i] = acc;
            }
            for (int j = 0; j < NV; j++)
            {
                double acc = 0.;
                for (int q = 0; q < NL; q++)
                    acc += Zk[q * NV + j] * c[q];
                g[j] = beta[size_t(k + 1) * NV + j] + acc;
            }
        }
        ow.assign(size_t(K) * NV, 0.);
        ol.assign(size_t(K - 1) * NL, 0.);
        {
            const double *Li = &Lif[size_t(K - 1) * 256];
            for (int i = 0; i < NV; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NV; q++)
                    acc += Li[q * NV + i] * av[size_t(K - 1) * NV + q];
                ow[size_t(K - 1) * NV + i] = acc;
            }
        }
        for (int k = K - 2; k >= 0; k--)
        {
            const double *Li = &Lif[size_t(k) * 256], *Yt = &Ytf[size_t(k) * 256], *Ti = &Tif[size_t(k) * 256],
                         *Zk = &Zf[size_t(k) * 256];
            const double *xn = &ow[size_t(k + 1) * NV];
            double t[NL], sv[NV];
            for (int i = 0; i < NL; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NV; q++)
                    acc += Zk[i * NV + q] * xn[q];
                t[i] = acc - cv[size_t(k) * NL + i];
            }
            double *lk = &ol[size_t(k) * NL];
            for (int i = 0; i < NL; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NL; q++)
                    acc += Ti[q * NL + i] * t[q];
                lk[i] = acc;
            }
            for (int j = 0; j < NV; j++)
            {
                double acc = 0.;
                for (int q = 0; q < NL; q++)
                    acc += Yt[j * NV + q] * lk[q];
                sv[j] = av[size_t(k) * NV + j] - acc;
            }
            double *x = &ow[size_t(k) * NV];
            for (int i = 0; i < NV; i++)
            {
                double acc = 0.;
                for (int q = 0; q < NV; q++)
                    acc += Li[q * NV + i] * sv[q];
                x[i] = acc;
            }
        }
    }

    // ---- generic reduced KKT solve ---------------------------------------------------------------
    // Solves   H dx + A' dy = bx ,  A dx = by   (H = L'W^-2 L over all cones but r3, r3 augmented)
    // with full-layout right-hand sides:
    //   bxw[K][16], bxd[K] (delta_k), bxnu, bxnub [K-1][14], bxs (sigma), bxds (delta_sigma), bxn1
    //   by [K-1][14]; rhs3 for the r3 row:  sum(dnub) - dn1 - (s3/z3) dz3 = rhs3
    struct Rhs
    {
        std::vector<double> w, d, nu, nub, y;
        double s = 0, ds = 0, n1 = 0, rhs3 = 0;
    };
    Rhs newRhs() const
    {
        using namespace sipm;
        Rhs r;
        r.w.assign(size_t(K) * NV, 0.);
        r.d.assign(K, 0.);
        r.nu.assign(size_t(K - 1) * NL, 0.);
        r.nub = r.nu;
        r.y = r.nu;
        return r;
    }
    void kktSolve(const Rhs &b)
    {
        using namespace sipm;
        // (1) n1 / r3
        dz3 = -b.n1;
        // (2) nu / nu_bound
        std::vector<double> beta(size_t(K) * NV), rho(size_t(K - 1) * NL), btn(size_t(K - 1) * NL),
            bnb(size_t(K - 1) * NL), dinv(size_t(K - 1) * NL);
        for (int k = 0; k < K - 1; k++)
            for (int i = 0; i < NL; i++)
            {
                const size_t o = size_t(k) * NL + i;
                const double d1 = dNu(s1[o], z1[o]), d2 = dNu(s2[o], z2[o]);
                dinv[o] = 1. / (d1 + d2);
                bnb[o] = b.nub[o] - dz3;
                btn[o] = b.nu[o] - qv[o] * bnb[o];
                rho[o] = b.y[o] + Einv[o] * btn[o];
            }
        // (3) delta_k
        for (int k = 0; k < K; k++)
            for (int j = 0; j < NV; j++)
                beta[size_t(k) * NV + j] =
                    (fm[k] & (1u << j)) ? 0. : b.w[size_t(k) * NV + j] - hdw[size_t(k) * NV + j] * b.d[k] / hdd[k];
        // (4) sigma / delta_sigma
        const double bts = b.s - Hsd * b.ds / Hdd;
        // (5) bordered block solve
        std::vector<double> vw, vl;
        blockSolve(beta, rho, vw, vl);
        double cv = 0.;
        for (int k = 0; k < K - 1; k++)
            for (int i = 0; i < NL; i++)
                cv += -P->S[size_t(k) * NX + i] * vl[size_t(k) * NL + i];
        dsig = (bts - cv) / schur_sig;
        for (size_t i = 0; i < vw.size(); i++)
            dw[i] = vw[i] - bcol_w[i] * dsig;
        for (size_t i = 0; i < vl.size(); i++)
            dlam[i] = vl[i] - bcol_l[i] * dsig;
        // (6) recover
        ddsg = (b.ds - Hsd * dsig) / Hdd;
        for (int k = 0; k < K; k++)
        {
            double acc = 0.;
            for (int j = 0; j < NV; j++)
                acc += hdw[size_t(k) * NV + j] * dw[size_t(k) * NV + j];
            ddl[k] = P->scvx ? 0. : (b.d[k] - acc) / hdd[k];
        }
        double sumnb = 0.;
        for (int k = 0; k < K - 1; k++)
            for (int i = 0; i < NL; i++)
            {
                const size_t o = size_t(k) * NL + i;
                dnu[o] = Einv[o] * (dlam[o] + btn[o]);
                dnub[o] = bnb[o] * dinv[o] - qv[o] * dnu[o];
                sumnb += dnub[o];
            }
        const double w3sq = identityScaling ? 1. : s3 / z3;
        dn1 = sumnb - w3sq * dz3 - b.rhs3;
    }

    // ---- main loop ---------------------------------------------------------------------------------
    int run(RQSocpOutput &out)
    {
        using namespace sipm;
        const double wtrx = P->w_trx;
        if (warm_start)
        {
            // primal point and multipliers of the previous solve; slacks re-evaluated on the new data and pushed
            // theta into the interior, duals likewise
            evalAllSaff(s, s1, s2, ss, s3, sc3);
            shiftToCone(s, s1, s2, ss, s3, sc3, warm_theta);
            shiftToCone(z, z1, z2, zs, z3, zc3, warm_theta);
        }
        else
        {
        // ---------- initialisation (ECOS init with W = I; two solves on one factorisation) ----------
        identityScaling = true;
        setIdentityScalings();
        if (!factor())
            return -2;
        // primal: min ||Gx-h||^2 s.t. Ax=b  ->  H dx + A'y = L'(-saff(x0)) ... derive: bx = G' r, r = saff(x0), G' = -L'
        {
            Rhs b = newRhs();
            std::vector<double> r(NS);
            for (int k = 0; k < K; k++)
            {
                saff(k, &w[size_t(k) * NV], dl[k], r.data());
                double gdl;
                LTmul(k, r.data(), &b.w[size_t(k) * NV], gdl);
                for (int j = 0; j < NV; j++)
                    b.w[size_t(k) * NV + j] = -b.w[size_t(k) * NV + j];
                b.d[k] = -gdl;
            }
            // segment rows r1: saff = nub - nu, r2: nub + nu (all zero at x0) ; r3: n1 - sum nub = 0
            // sigma lp: saff = sigma - 0.001 = -0.001 ; cone: (0.5, 0.5, -sigbar) at sigma=dsg=0
            // G' r for sigma: L rows: lp: +1 ; cone third: +1  => L'r = r_lp + r_c[2]; bx = -L'r
            b.s = -((sig - 0.001) + (sig - P->sigbar));
            b.ds = -(0.5 * (0.5 + 0.5 * dsg) - 0.5 * (0.5 - 0.5 * dsg));
            b.rhs3 = -(n1 - 0.); // r3 row: G3 dx - dz3 = h3 - G3 x0 ... = saff3(x0)  (with dz3 = 0)
            // by = -ry(x0)
            for (int k = 0; k < K - 1; k++)
            {
                double res[NL];
                dynRes(k, w.data(), nu.data(), sig, res);
                for (int i = 0; i < NL; i++)
                    b.y[size_t(k) * NL + i] = -res[i];
            }
            // r3 row in "augmented" form:  G3 dx - dz3 = r3_aff  with G3 dx = sum dnub - dn1, unknown dz3 = -bx_n1 = 0
            // => dn1 = sum dnub - r3_aff ; kktSolve computes dn1 = sum - w3sq*dz3 - rhs3  => rhs3 = r3_aff = n1 - sum nub
            b.rhs3 = n1;
            kktSolve(b);
            applyPrimalStep(1.);
            // s = bring2cone(saff(x))
            evalAllSaff(s, s1, s2, ss, s3, sc3);
            bring2cone(s, s1, s2, ss, s3, sc3);
        }
        // dual: K [x';y;z] = [-c;0;0]  ->  H x' + A'y = -c ; z = G x' = -L x'
        {
            Rhs b = newRhs();
            for (int k = 0; k < K; k++)
                b.d[k] = -wtrx;
            b.s = -P->w_t;
            b.ds = -P->w_trt;
            b.n1 = -P->w_vc;
            b.rhs3 = 0.;
            kktSolve(b);
            lam = dlam;
            // z = -L(dx)
            std::vector<double> t(NS);
            for (int k = 0; k < K; k++)
            {
                Lmul(k, &dw[size_t(k) * NV], ddl[k], t.data());
                for (int i = 0; i < NS; i++)
                    z[size_t(k) * NS + i] = -t[i];
            }
            for (int k = 0; k < K - 1; k++)
                for (int i = 0; i < NL; i++)
                {
                    const size_t o = size_t(k) * NL + i;
                    z1[o] = -(dnub[o] - dnu[o]);
                    z2[o] = -(dnub[o] + dnu[o]);
                }
            zs = -dsig;
            z3 = dz3;
            zc3[0] = -0.5 * ddsg;
            zc3[1] = 0.5 * ddsg;
            zc3[2] = -dsig;
            bring2cone(z, z1, z2, zs, z3, zc3);
        }
        }
        identityScaling = false;

        // norms of problem data for the termination test (ECOS-style scaling)
        const double resx0 = std::max(1., std::sqrt(K * wtrx * wtrx + sq(P->w_t) + sq(P->w_trt) + sq(P->w_vc)));
        double resy0, resz0;
        {
            // b_eff = -ry at free=0 ; h_eff = saff at free=0
            std::vector<double> w0(w), nu0(nu.size(), 0.);
            for (int k = 0; k < K; k++)
                for (int j = 0; j < NV; j++)
                    if (!(fm[k] & (1u << j)))
                        w0[size_t(k) * NV + j] = 0.;
            double nb = 0., nh = 0.;
            for (int k = 0; k < K - 1; k++)
            {
                double res[NL];
                dynRes(k, w0.data(), nu0.data(), 0., res);
                for (int i = 0; i < NL; i++)
                    nb += res[i] * res[i];
            }
            std::vector<double> r(NS);
            for (int k = 0; k < K; k++)
            {
                // saff with free variables zero
                const std::vector<double> wsave(w.begin() + size_t(k) * NV, w.begin() + size_t(k + 1) * NV);
                saff(k, &w0[size_t(k) * NV], 0., r.data());
                for (int i = 0; i < NS; i++)
                    nh += r[i] * r[i];
            }
            nh += 0.001 * 0.001 + 0.25 + 0.25 + sq(P->sigbar);
            resy0 = std::max(1., std::sqrt(nb));
            resz0 = std::max(1., std::sqrt(nh));
        }

        std::vector<double> rxw(size_t(K) * NV), rxd(K), rxnu(size_t(K - 1) * NL), rxnub(size_t(K - 1) * NL),
            ry(size_t(K - 1) * NL), rz(size_t(K) * NS), rz1(size_t(K - 1) * NL), rz2(size_t(K - 1) * NL);
        double rxs, rxds, rxn1, rzs, rz3, rzc[3];
        std::vector<double> sa(size_t(K) * NS), sa1(size_t(K - 1) * NL), sa2(size_t(K - 1) * NL);
        double sas, sa3, sac[3];
        std::vector<double> tz(size_t(K) * NS), tz1(size_t(K - 1) * NL), tz2(size_t(K - 1) * NL);
        double tzs, tzc[3];
        std::vector<double> lamS(size_t(K) * NS), dsS(size_t(K) * NS), dzS(size_t(K) * NS); // scaled quantities
        double lamC[3], dsC[3], dzC[3];

        // ECOS-style safeguarding: the last iterate that met the reduced tolerances is kept; if the residuals then
        // explode (pres > 500 x previous, negative gap, NaN) that iterate is restored and returned
        std::vector<double> bk_w, bk_dl;
        double bk_sig = 0., bk_dsg = 0., bk_n1 = 0., pres_prev = 0.;
        bool bk_valid = false;
        auto restoreBest = [&]() {
            w = bk_w;
            dl = bk_dl;
            sig = bk_sig;
            dsg = bk_dsg;
            n1 = bk_n1;
        };
        for (int iter = 0;; iter++)
        {
            // ---------- residuals ----------
            // rz = s - saff(x)
            evalAllSaff(sa, sa1, sa2, sas, sa3, sac);
            for (size_t i = 0; i < rz.size(); i++)
                rz[i] = s[i] - sa[i];
            for (size_t i = 0; i < rz1.size(); i++)
            {
                rz1[i] = s1[i] - sa1[i];
                rz2[i] = s2[i] - sa2[i];
            }
            rzs = ss - sas;
            rz3 = s3 - sa3;
            for (int i = 0; i < 3; i++)
                rzc[i] = sc3[i] - sac[i];
            // ry
            for (int k = 0; k < K - 1; k++)
                dynRes(k, w.data(), nu.data(), sig, &ry[size_t(k) * NL]);
            // rx = c + A'y - L'z
            for (int k = 0; k < K; k++)
            {
                double gw[NV], gdl;
                LTmul(k, &z[size_t(k) * NS], gw, gdl);
                rxd[k] = P->scvx ? 0. : wtrx - gdl;
                double *r = &rxw[size_t(k) * NV];
                for (int j = 0; j < NV; j++)
                    r[j] = -gw[j];
                if (k < K - 1)
                {
                    double M[NL * NV];
                    buildM(k, M);
                    for (int i = 0; i < NL; i++)
                        for (int j = 0; j < NV; j++)
                            r[j] += M[i * NV + j] * lam[size_t(k) * NL + i];
                }
                if (k > 0)
                {
                    double N[NL * NV];
                    buildN(k - 1, N);
                    for (int i = 0; i < NL; i++)
                        for (int j = 0; j < NV; j++)
                            r[j] += N[i * NV + j] * lam[size_t(k - 1) * NL + i];
                }
            }
            rxs = P->w_t - zs - zc3[2];
            rxds = P->w_trt - 0.5 * zc3[0] + 0.5 * zc3[1];
            rxn1 = P->w_vc - z3;
            for (int k = 0; k < K - 1; k++)
                for (int i = 0; i < NL; i++)
                {
                    const size_t o = size_t(k) * NL + i;
                    rxnu[o] = -lam[o] + z1[o] - z2[o];
                    rxnub[o] = -z1[o] - z2[o] + z3;
                    rxs -= P->S[size_t(k) * NX + i] * lam[o];
                }
            // ---------- statistics ----------
            double gap = 0., nrx = 0., nry = 0., nrz = 0., nxx = 0., nyy = 0., nzz = 0., nss = 0.;
            for (size_t i = 0; i < s.size(); i++)
            {
                gap += s[i] * z[i];
                nrz += rz[i] * rz[i];
                nzz += z[i] * z[i];
                nss += s[i] * s[i];
            }
            for (size_t i = 0; i < s1.size(); i++)
            {
                gap += s1[i] * z1[i] + s2[i] * z2[i];
                nrz += rz1[i] * rz1[i] + rz2[i] * rz2[i];
                nzz += z1[i] * z1[i] + z2[i] * z2[i];
                nss += s1[i] * s1[i] + s2[i] * s2[i];
                nry += ry[i] * ry[i];
                nyy += lam[i] * lam[i];
                nrx += rxnu[i] * rxnu[i] + rxnub[i] * rxnub[i];
                nxx += nu[i] * nu[i] + nub[i] * nub[i];
            }
            gap += ss * zs + s3 * z3;
            nrz += rzs * rzs + rz3 * rz3;
            nzz += zs * zs + z3 * z3;
            nss += ss * ss + s3 * s3;
            for (int i = 0; i < 3; i++)
            {
                gap += sc3[i] * zc3[i];
                nrz += rzc[i] * rzc[i];
                nzz += zc3[i] * zc3[i];
                nss += sc3[i] * sc3[i];
            }
            for (int k = 0; k < K; k++)
            {
                nrx += rxd[k] * rxd[k];
                nxx += dl[k] * dl[k];
                for (int j = 0; j < NV; j++)
                    if (!(fm[k] & (1u << j)))
                    {
                        nrx += sq(rxw[size_t(k) * NV + j]);
                        nxx += sq(w[size_t(k) * NV + j]);
                    }
            }
            nrx += rxs * rxs + rxds * rxds + rxn1 * rxn1;
            nxx += sig * sig + dsg * dsg + n1 * n1;
            const double mu = gap / D;
            double pcost = P->w_t * sig + P->w_trt * dsg + P->w_vc * n1;
            for (int k = 0; k < K; k++)
                pcost += wtrx * dl[k];
            const double nx_ = std::sqrt(nxx), ny_ = std::sqrt(nyy), nz_ = std::sqrt(nzz), ns_ = std::sqrt(nss);
            const double pres = std::max(std::sqrt(nry) / std::max(resy0 + nx_, 1.), std::sqrt(nrz) / std::max(resz0 + nx_ + ns_, 1.));
            const double dres = std::sqrt(nrx) / std::max(resx0 + ny_ + nz_, 1.);
            const double relgap = gap / std::max(std::fabs(pcost), 1e-300);
            out.iters = iter;
            out.pres = pres;
            out.dres = dres;
            out.gap = gap;
            out.pcost = pcost;
            if (opt.verbose)
                std::printf("%3d  pcost %+.8e gap %.2e pres %.2e dres %.2e mu %.2e sigma %.6f n1 %.3e\n", iter, pcost, gap,
                            pres, dres, mu, sig, n1);
            // (1e30: csrc/ipm_solve.h IPM_BLOWN -- a blown-up iterate has pres ~ 0 in these RELATIVE measures and would pass as optimal)
            // (round 6: two-sided in the gap, and a gap below -1e-6 = IPM_NEG_GAP is broken with or without a fall-back: the iterate left the cone)
            if (!std::isfinite(pres) || !std::isfinite(dres) || !std::isfinite(gap) || std::fabs(pcost) > 1e30 || std::fabs(gap) > 1e30 || gap < -1e-6 ||
                (bk_valid && iter > 0 && (pres > 500. * pres_prev || gap < 0.)))
            {
                if (!bk_valid)
                    return -2;
                restoreBest();
                restored_best = true; // slacks / duals are those of the broken iterate: no warm start from here
                return 0;
            }
            pres_prev = pres;
            if (pres < opt.feastol && dres < opt.feastol && (gap < opt.abstol || relgap < opt.reltol))
                return 0;
            // ECOS's reduced-accuracy exit (feastol_inacc 1e-4, abstol_inacc / reltol_inacc 5e-5): when the iteration
            // limit or a numerical breakdown is hit at an iterate that already satisfies the relaxed tolerances, ECOS
            // returns it as "close to optimal" instead of failing
            const bool inacc_ok = pres < 1e-4 && dres < 1e-4 && (gap < 5e-5 || relgap < 5e-5);
            if (inacc_ok)
            {
                bk_w = w;
                bk_dl = dl;
                bk_sig = sig;
                bk_dsg = dsg;
                bk_n1 = n1;
                bk_valid = true;
            }
            if (iter >= opt.maxit)
                return inacc_ok ? 0 : -1;

            // ---------- scalings, factorisation ----------
            if (!updateScalings())
            {
                last_fail = 1;
                return inacc_ok ? 0 : -2;
            }
            if (!factor())
            {
                if (opt.verbose)
                    std::printf("factor failed: code %d\n", last_fail);
                return inacc_ok ? 0 : -2;
            }
            // lambda = W z (scaled variable), per cone
            for (int k = 0; k < K; k++)
            {
                for (int c = 0; c < NCONE; c++)
                    if (act[k] & (1u << c))
                        applyW(scal[size_t(k) * NCONE + c], cone_dim[c], &z[size_t(k) * NS + cone_off[c]],
                               &lamS[size_t(k) * NS + cone_off[c]]);
            }
            applyW(scsig, 3, zc3, lamC);

            double sigma_c = 0., alpha = 1., alpha_pr = 1., alpha_du = 1.;
            for (int pass = 0; pass < 2; pass++)
            {
                const double om = 1. - sigma_c; // residual scaling
                // ---------- t = W^-2 rz' + W^-1 (lambda \ ds) ----------
                for (int k = 0; k < K; k++)
                {
                    for (int c = 0; c < NCONE; c++)
                    {
                        const int o = int(size_t(k) * NS) + cone_off[c], d = cone_dim[c];
                        if (!(act[k] & (1u << c)))
                        {
                            for (int i = 0; i < d; i++)
                                tz[o + i] = 0.;
                            continue;
                        }
                        const Scaling &sc = scal[size_t(k) * NCONE + c];
                        double a[17], b2[17], dsv[17], u[17];
                        for (int i = 0; i < d; i++)
                            a[i] = om * rz[o + i];
                        applyWinv2(sc, d, a, b2);
                        if (pass == 0)
                        {
                            // lambda\(-lambda o lambda) = -lambda ; W^-1(-lambda) = -z
                            for (int i = 0; i < d; i++)
                                tz[o + i] = b2[i] - z[o + i];
                        }
                        else
                        {
                            conicProduct(d, &dsS[o], &dzS[o], dsv);
                            for (int i = 0; i < d; i++)
                                dsv[i] = -dsv[i];
                            dsv[0] += sigma_c * mu;
                            conicDivision(d, &lamS[o], dsv, u);
                            for (int i = 0; i < d; i++)
                                u[i] -= lamS[o + i];
                            applyWinv(sc, d, u, a);
                            for (int i = 0; i < d; i++)
                                tz[o + i] = b2[i] + a[i];
                        }
                    }
                    for (int which = 0; which < 2; which++)
                    {
                        const int o = int(size_t(k) * NS) + (which ? L2 : L1);
                        if (!(act[k] & (1u << (6 + which))))
                        {
                            tz[o] = 0.;
                            continue;
                        }
                        const double corr = pass ? (sigma_c * mu - ds[o] * dz[o]) / s[o] : 0.;
                        tz[o] = (z[o] / s[o]) * om * rz[o] - z[o] + corr;
                    }
                }
                for (size_t o = 0; o < s1.size(); o++)
                {
                    const double c1 = pass ? (sigma_c * mu - ds1[o] * dz1[o]) / s1[o] : 0.;
                    const double c2 = pass ? (sigma_c * mu - ds2[o] * dz2[o]) / s2[o] : 0.;
                    tz1[o] = dNu(s1[o], z1[o]) * om * rz1[o] - z1[o] + c1;
                    tz2[o] = dNu(s2[o], z2[o]) * om * rz2[o] - z2[o] + c2;
                }
                tzs = (zs / ss) * om * rzs - zs + (pass ? (sigma_c * mu - dss * dzs) / ss : 0.);
                {
                    double a[3], b2[3], dsv[3], u[3];
                    for (int i = 0; i < 3; i++)
                        a[i] = om * rzc[i];
                    applyWinv2(scsig, 3, a, b2);
                    if (pass == 0)
                        for (int i = 0; i < 3; i++)
                            tzc[i] = b2[i] - zc3[i];
                    else
                    {
                        conicProduct(3, dsC, dzC, dsv);
                        for (int i = 0; i < 3; i++)
                            dsv[i] = -dsv[i];
                        dsv[0] += sigma_c * mu;
                        conicDivision(3, lamC, dsv, u);
                        for (int i = 0; i < 3; i++)
                            u[i] -= lamC[i];
                        applyWinv(scsig, 3, u, a);
                        for (int i = 0; i < 3; i++)
                            tzc[i] = b2[i] + a[i];
                    }
                }
                // r3 (augmented): rhs3 = -rz3' - ds3/z3 with ds3 = -s3 z3 (+ corrections)
                const double ds3v = -s3 * z3 + (pass ? (sigma_c * mu - ds3 * dz3) : 0.);
                // ---------- right-hand side  bx = -rx' + L't ----------
                Rhs b = newRhs();
                for (int k = 0; k < K; k++)
                {
                    double gw[NV], gdl;
                    LTmul(k, &tz[size_t(k) * NS], gw, gdl);
                    for (int j = 0; j < NV; j++)
                        b.w[size_t(k) * NV + j] = -om * rxw[size_t(k) * NV + j] + gw[j];
                    b.d[k] = -om * rxd[k] + gdl;
                }
                for (size_t o = 0; o < s1.size(); o++)
                {
                    // L rows: r1 = nub - nu: (nu:-1, nub:+1) ; r2 = nub + nu: (nu:+1, nub:+1)
                    b.nu[o] = -om * rxnu[o] + (-tz1[o] + tz2[o]);
                    b.nub[o] = -om * rxnub[o] + (tz1[o] + tz2[o]);
                    b.y[o] = -om * ry[o];
                }
                b.s = -om * rxs + tzs + tzc[2];
                b.ds = -om * rxds + 0.5 * tzc[0] - 0.5 * tzc[1];
                b.n1 = -om * rxn1;
                b.rhs3 = -om * rz3 - ds3v / z3;
                kktSolve(b);
                {
                    // a non-finite Newton direction (breakdown of the factorisation near the end of the path) must not be
                    // applied: same reduced-accuracy exit as for the other numerical failures
                    double chk = dsig * 0. + ddsg * 0. + dn1 * 0.;
                    for (double v : dw)
                        chk += v * 0.;
                    for (double v : dlam)
                        chk += v * 0.;
                    if (!(chk == 0.))
                        return inacc_ok ? 0 : -2;
                }
                // ---------- dz = -W^-2 L dx + t ;  ds = -rz' + L dx ----------
                double ainv = 0., ainv_p = 0., ainv_d = 0.; // 1 / alpha_max: common, of the slack directions, of the multiplier directions
                for (int k = 0; k < K; k++)
                {
                    double Ld[NS];
                    Lmul(k, &dw[size_t(k) * NV], ddl[k], Ld);
                    for (int c = 0; c < NCONE; c++)
                    {
                        const int o = int(size_t(k) * NS) + cone_off[c], d = cone_dim[c];
                        if (!(act[k] & (1u << c)))
                            continue;
                        const Scaling &sc = scal[size_t(k) * NCONE + c];
                        double a[17];
                        applyWinv2(sc, d, &Ld[cone_off[c]], a);
                        for (int i = 0; i < d; i++)
                        {
                            dz[o + i] = -a[i] + tz[o + i];
                            ds[o + i] = -om * rz[o + i] + Ld[cone_off[c] + i];
                        }
                        applyWinv(sc, d, &ds[o], &dsS[o]);
                        applyW(sc, d, &dz[o], &dzS[o]);
                        ainv_p = std::max(ainv_p, stepInv(d, &lamS[o], &dsS[o]));
                        ainv_d = std::max(ainv_d, stepInv(d, &lamS[o], &dzS[o]));
                    }
                    for (int which = 0; which < 2; which++)
                    {
                        const int li = which ? L2 : L1;
                        const int o = int(size_t(k) * NS) + li;
                        if (!(act[k] & (1u << (6 + which))))
                            continue;
                        dz[o] = -(z[o] / s[o]) * Ld[li] + tz[o];
                        ds[o] = -om * rz[o] + Ld[li];
                        ainv_p = std::max(ainv_p, -ds[o] / s[o]);
                        ainv_d = std::max(ainv_d, -dz[o] / z[o]);
                    }
                }
                for (size_t o = 0; o < s1.size(); o++)
                {
                    const double L1v = dnub[o] - dnu[o], L2v = dnub[o] + dnu[o];
                    dz1[o] = -dNu(s1[o], z1[o]) * L1v + tz1[o];
                    ds1[o] = -om * rz1[o] + L1v;
                    dz2[o] = -dNu(s2[o], z2[o]) * L2v + tz2[o];
                    ds2[o] = -om * rz2[o] + L2v;
                    ainv_p = std::max(ainv_p, -ds1[o] / s1[o]);
                    ainv_d = std::max(ainv_d, -dz1[o] / z1[o]);
                    ainv_p = std::max(ainv_p, -ds2[o] / s2[o]);
                    ainv_d = std::max(ainv_d, -dz2[o] / z2[o]);
                }
                dzs = -(zs / ss) * dsig + tzs;
                dss = -om * rzs + dsig;
                ainv_p = std::max(ainv_p, -dss / ss);
                ainv_d = std::max(ainv_d, -dzs / zs);
                // r3: saff = n1 - sum nub
                {
                    double sumnb = 0.;
                    for (double v : dnub)
                        sumnb += v;
                    ds3 = -om * rz3 + (dn1 - sumnb);
                    ainv_p = std::max(ainv_p, -ds3 / s3);
                    ainv_d = std::max(ainv_d, -dz3 / z3);
                }
                {
                    const double Ld[3] = {0.5 * ddsg, -0.5 * ddsg, dsig};
                    double a[3];
                    applyWinv2(scsig, 3, Ld, a);
                    for (int i = 0; i < 3; i++)
                    {
                        dzc3[i] = -a[i] + tzc[i];
                        dsc3[i] = -om * rzc[i] + Ld[i];
                    }
                    applyWinv(scsig, 3, dsc3, dsC);
                    applyW(scsig, 3, dzc3, dzC);
                    ainv_p = std::max(ainv_p, stepInv(3, lamC, dsC));
                    ainv_d = std::max(ainv_d, stepInv(3, lamC, dzC));
                    ainv = std::max(ainv_p, ainv_d);
                }
                if (pass == 0)
                {
                    const double alpha_a = ainv > 0. ? std::min(1. / ainv, 1.) : 1.;
                    sigma_c = (1. - alpha_a) * (1. - alpha_a) * (1. - alpha_a);
                    sigma_c = std::min(1., std::max(1e-4, sigma_c));
                }
                else
                {
                    alpha = ainv > 0. ? std::min(opt.gamma / ainv, 1.) : 1.;
                    alpha = std::min(alpha, 0.999);
                    alpha = std::max(alpha, 1e-8);
                    alpha_pr = alpha_du = alpha;
                    if (opt.split_steps)
                    {
                        alpha_pr = ainv_p > 0. ? std::min(opt.gamma / ainv_p, 1.) : 1.;
                        alpha_du = ainv_d > 0. ? std::min(opt.gamma / ainv_d, 1.) : 1.;
                        alpha_pr = std::max(std::min(alpha_pr, 0.999), 1e-8);
                        alpha_du = std::max(std::min(alpha_du, 0.999), 1e-8);
                    }
                }
            }
            // ---------- update ----------
            applyPrimalStep(alpha_pr);
            for (size_t i = 0; i < lam.size(); i++)
                lam[i] += alpha_du * dlam[i];
            for (size_t i = 0; i < s.size(); i++)
            {
                s[i] += alpha_pr * ds[i];
                z[i] += alpha_du * dz[i];
            }
            for (size_t i = 0; i < s1.size(); i++)
            {
                s1[i] += alpha_pr * ds1[i];
                z1[i] += alpha_du * dz1[i];
                s2[i] += alpha_pr * ds2[i];
                z2[i] += alpha_du * dz2[i];
            }
            ss += alpha_pr * dss;
            zs += alpha_du * dzs;
            s3 += alpha_pr * ds3;
            z3 += alpha_du * dz3;
            for (int i = 0; i < 3; i++)
            {
                sc3[i] += alpha_pr * dsc3[i];
                zc3[i] += alpha_du * dzc3[i];
            }
        }
    }
    void applyPrimalStep(double alpha)
    {
        using namespace sipm;
        for (int k = 0; k < K; k++)
        {
            for (int j = 0; j < NV; j++)
                if (!(fm[k] & (1u << j)))
                    w[size_t(k) * NV + j] += alpha * dw[size_t(k) * NV + j];
            dl[k] += alpha * ddl[k];
        }
        for (size_t i = 0; i < nu.size(); i++)
        {
            nu[i] += alpha * dnu[i];
            nub[i] += alpha * dnub[i];
        }
        sig += alpha * dsig;
        dsg += alpha * ddsg;
        n1 += alpha * dn1;
    }

    void evalAllSaff(std::vector<double> &o, std::vector<double> &o1, std::vector<double> &o2, double &os, double &o3,
                     double *oc) const
    {
        using namespace sipm;
        for (int k = 0; k < K; k++)
            saff(k, &w[size_t(k) * NV], dl[k], &o[size_t(k) * NS]);
        double sumnb = 0.;
        for (size_t i = 0; i < nu.size(); i++)
        {
            o1[i] = nub[i] - nu[i];
            o2[i] = nub[i] + nu[i];
            sumnb += nub[i];
        }
        os = sig - 0.001;
        o3 = n1 - sumnb;
        oc[0] = 0.5 + 0.5 * dsg;
        oc[1] = 0.5 - 0.5 * dsg;
        oc[2] = sig - P->sigbar;
    }

    // warm start: v += (theta + max violation) e over the whole product cone
    void shiftToCone(std::vector<double> &v, std::vector<double> &v1, std::vector<double> &v2, double &vs, double &v3,
                     double *vc, double theta) const
    {
        using namespace sipm;
        double alpha = 0.;
        auto lp = [&](double r) {
            if (-r > alpha)
                alpha = -r;
        };
        auto soc = [&](const double *r, int d) {
            double nrm = 0.;
            for (int i = 1; i < d; i++)
                nrm += r[i] * r[i];
            lp(r[0] - std::sqrt(nrm));
        };
        for (int k = 0; k < K; k++)
        {
            for (int c = 0; c < NCONE; c++)
                if (act[k] & (1u << c))
                    soc(&v[size_t(k) * NS + cone_off[c]], cone_dim[c]);
            if (act[k] & 64u)
                lp(v[size_t(k) * NS + L1]);
            if (act[k] & 128u)
                lp(v[size_t(k) * NS + L2]);
        }
        for (size_t i = 0; i < v1.size(); i++)
        {
            lp(v1[i]);
            lp(v2[i]);
        }
        lp(vs);
        lp(v3);
        soc(vc, 3);
        alpha += theta;
        for (int k = 0; k < K; k++)
        {
            for (int c = 0; c < NCONE; c++)
                if (act[k] & (1u << c))
                    v[size_t(k) * NS + cone_off[c]] += alpha;
            if (act[k] & 64u)
                v[size_t(k) * NS + L1] += alpha;
            if (act[k] & 128u)
                v[size_t(k) * NS + L2] += alpha;
        }
        for (size_t i = 0; i < v1.size(); i++)
        {
            v1[i] += alpha;
            v2[i] += alpha;
        }
        vs += alpha;
        v3 += alpha;
        vc[0] += alpha;
    }
    // ECOS bring2cone over the whole product cone
    void bring2cone(std::vector<double> &v, std::vector<double> &v1, std::vector<double> &v2, double &vs, double &v3,
                    double *vc) const
    {
        using namespace sipm;
        double alpha = -opt.gamma;
        auto lp = [&](double r) {
            if (r <= 0. && -r > alpha)
                alpha = -r;
        };
        auto soc = [&](const double *r, int d) {
            double nrm = 0.;
            for (int i = 1; i < d; i++)
                nrm += r[i] * r[i];
            const double cres = r[0] - std::sqrt(nrm);
            if (cres <= 0. && -cres > alpha)
                alpha = -cres;
        };
        for (int k = 0; k < K; k++)
        {
            for (int c = 0; c < NCONE; c++)
                if (act[k] & (1u << c))
                    soc(&v[size_t(k) * NS + cone_off[c]], cone_dim[c]);
            if (act[k] & 64u)
                lp(v[size_t(k) * NS + L1]);
            if (act[k] & 128u)
                lp(v[size_t(k) * NS + L2]);
        }
        for (size_t i = 0; i < v1.size(); i++)
        {
            lp(v1[i]);
            lp(v2[i]);
        }
        lp(vs);
        lp(v3);
        soc(vc, 3);
        alpha += 1.;
        for (int k = 0; k < K; k++)
        {
            for (int c = 0; c < NCONE; c++)
                if (act[k] & (1u << c))
                    v[size_t(k) * NS + cone_off[c]] += alpha;
            if (act[k] & 64u)
                v[size_t(k) * NS + L1] += alpha;
            if (act[k] & 128u)
                v[size_t(k) * NS + L2] += alpha;
        }
        for (size_t i = 0; i < v1.size(); i++)
        {
            v1[i] += alpha;
            v2[i] += alpha;
        }
        vs += alpha;
        v3 += alpha;
        vc[0] += alpha;
    }
};

} // namespace oracle
