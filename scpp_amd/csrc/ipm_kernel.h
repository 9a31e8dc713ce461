// Batched structure-exploiting interior-point solver for the RocketQuat SC sub-problem.
// Replaces, for B problem instances at once, the reference's per-iteration
//   solver->solve(false)      scpp_core/src/SCAlgorithm.cpp:78   (Epigraph -> ECOS)
// on the problem of buildSCProblem (scpp_core/src/SCProblem.cpp:6-138) +
// RocketQuat::addApplicationConstraints (scpp_models/src/rocketQuat.cpp:70-144), followed by
// readSolution + the convergence/weight logic of SCAlgorithm::iterate (SCAlgorithm.cpp:100-131).
//
// Formulation (scalar twin with the derivation: oracle/structured_ipm.hpp):
//   * presolve: x_0, the fixed final-state components, U[{0,1},K-1], X[13,:] and U[3,:] are constants;
//     16 stage variables w_k = (x_k[0..12], u_k[0..2]) remain, plus delta_k, nu_k, nu_bound_k, sigma,
//     delta_sigma, norm1_nu;
//   * primal-dual Mehrotra predictor-corrector with Nesterov-Todd scaling (the ECOS scheme), no
//     self-dual embedding (virtual control makes every sub-problem feasible);
//   * per IPM iteration ONE factorisation of the reduced KKT system: nu, nu_bound, norm1_nu, delta_k,
//     delta_sigma eliminated in closed form, then a block-tridiagonal quasi-definite system
//       [H_k M_k'; M_k -E_k^-1] ... coupled by N_k, with sigma as a one-column border,
//     factorised stage by stage with dense 16x16 Cholesky tiles held in LDS (FP64 MFMA
//     v_mfma_f64_16x16x4_f64 for the Z'Z / YY' tile products).
//
// Mapping: ONE 64-lane wavefront (one workgroup) per problem instance. Element-wise phases run with
// lane == stage/segment index (K <= 64); the factorisation / substitution sweeps are sequential over
// stages with the 64 lanes cooperating on the 16x16 tiles.  All per-instance state lives in an HBM
// workspace (layout below, ~0.66 MB per instance for K = 50).
#pragma once
#include "common.h"
#include "cone_math.h"

namespace scpp
{
namespace ipm
{

constexpr int NX = 14, NU = 4, NV = 16, NS = 35, NL = 14, NCONE = 6;
constexpr int C1 = 0, C2 = 17, C3 = 20, C4 = 23, C5 = 26, C6 = 30, L1 = 33, L2 = 34;
__device__ inline int coneOff(int c) { return c == 0 ? 0 : c == 1 ? 17 : c == 2 ? 20 : c == 3 ? 23 : c == 4 ? 26 : 30; }
__device__ inline int coneDim(int c) { return c == 0 ? 17 : c == 4 ? 4 : 3; }

// ---- per-instance parameter block (doubles) ----
enum InstPar
{
    IP_XINIT = 0,   // [14] nondimensional
    IP_XFINAL = 14, // [14]
    IP_GS = 28,
    IP_TILT,
    IP_WMAX,
    IP_TMIN,
    IP_TMAX,
    IP_GIM,
    IP_MDRY,
    IP_WT = 35,
    IP_WTRT,
    IP_WTRX,
    IP_WVC,
    IP_PAR = 39, // [10] flow-map parameters
    IP_MSCALE = 49,
    IP_RSCALE,
    IP_FINALTIME,
    IP_N = 56
};

// ---- stage record (doubles) ----
enum StageField
{
    F_W = 0,      // [16] stage variables
    F_DL = 16,    // delta_k
    F_DW = 17,    // [16]
    F_DDL = 33,
    F_WBAR = 34,  // [16] trust-region centre
    F_UHAT = 50,  // [3]
    F_HDD = 53,
    F_HDW = 54,   // [16]
    F_RXW = 70,   // [16]
    F_RXD = 86,
    F_BETA = 87,  // [16]
    F_BCW = 103,  // [16] border column (w part)
    F_VW = 119,   // [16] block-solve output
    F_AV = 135,   // [16] forward-sweep intermediate
    F_S = 151,    // [35]
    F_Z = 186,
    F_DS = 221,
    F_DZ = 256,
    F_RZ = 291,
    F_TZ = 326,
    F_LS = 361,   // lambda (scaled)
    F_DSS = 396,  // W^-1 ds
    F_DZS = 431,  // W dz
    F_ETA = 466,  // [6]
    F_WB = 472,   // [33] wbar of the 6 cones, same offsets as the slack layout
    F_BXW = 505,  // [16] right-hand side (w part)
    F_BXD = 521,
    STREC = 528
};
// ---- segment record: 33 fields of 14 doubles ----
enum SegField
{
    G_NU = 0,
    G_NUB,
    G_S1,
    G_Z1,
    G_S2,
    G_Z2,
    G_DNU,
    G_DNUB,
    G_DS1,
    G_DZ1,
    G_DS2,
    G_DZ2,
    G_LAM,
    G_DLAM,
    G_RY,
    G_RXNU,
    G_RXNUB,
    G_RZ1,
    G_RZ2,
    G_TZ1,
    G_TZ2,
    G_EINV,
    G_QV,
    G_RHO,
    G_BTN,
    G_BNB,
    G_DINV,
    G_BCL,
    G_VL,
    G_CV,
    G_BXNU,
    G_BXNUB,
    G_BY,
    G_NFIELDS
};
constexpr int SEGREC = G_NFIELDS * NL; // 462
constexpr int FACREC = 4 * 256;        // L, Y, T, Z tiles (16x16 row-major each)

__host__ __device__ inline size_t workspaceDoubles(int K) { return size_t(K) * STREC + size_t(K) * SEGREC + size_t(K) * FACREC; }

struct Settings
{
    double feastol, abstol, reltol, gamma;
    int maxit;
    int use_mfma;
};

// everything a wavefront needs to know about its instance
struct Ctx
{
    int K, lane;
    double *st;  // [K][STREC]
    double *sg;  // [K][SEGREC]
    double *fac; // [K][FACREC]
    const double *A, *B, *C, *S, *Z; // dd of this instance
    const double *ip;                // instance parameters
};

__device__ inline unsigned fixedMask(int k, int K)
{
    if (k == 0)
        return 0x1FFFu;
    if (k == K - 1)
        return (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 11) |
               (1u << 12) | (1u << 13) | (1u << 14);
    return 0u;
}
// bits 0..5 cones C1..C6, bit 6 mass LP, bit 7 min-thrust LP
__device__ inline unsigned activeMask(int k, int K)
{
    if (k == 0)
        return 0xFFu & ~((1u << 1) | (1u << 2) | (1u << 3) | (1u << 6));
    if (k == K - 1)
        return 0xFFu & ~((1u << 1) | (1u << 2) | (1u << 3));
    return 0xFFu;
}

__device__ inline void maskInactive(unsigned act, double *v)
{
    for (int c = 0; c < NCONE; c++)
        if (!(act & (1u << c)))
            for (int i = 0; i < coneDim(c); i++)
                v[coneOff(c) + i] = 0.;
    if (!(act & 64u))
        v[L1] = 0.;
    if (!(act & 128u))
        v[L2] = 0.;
}
// affine slack h - Gx of one stage
__device__ inline void saff(const double *ip, unsigned act, const double *wk, double dlk, const double *wb, const double *uh,
                            double *out)
{
    out[0] = dlk;
    for (int j = 0; j < NV; j++)
        out[1 + j] = wb[j] - wk[j];
    out[17] = ip[IP_GS] * wk[3];
    out[18] = wk[1];
    out[19] = wk[2];
    out[20] = ip[IP_TILT];
    out[21] = wk[8];
    out[22] = wk[9];
    out[23] = ip[IP_WMAX];
    out[24] = wk[11];
    out[25] = wk[12];
    out[26] = ip[IP_TMAX];
    out[27] = wk[13];
    out[28] = wk[14];
    out[29] = wk[15];
    out[30] = ip[IP_GIM] * wk[15];
    out[31] = wk[13];
    out[32] = wk[14];
    out[33] = wk[0] - ip[IP_MDRY];
    out[34] = uh[0] * wk[13] + uh[1] * wk[14] + uh[2] * wk[15] - ip[IP_TMIN];
    maskInactive(act, out);
}
// linear part of saff
__device__ inline void Lmul(const double *ip, unsigned act, const double *dwk, double ddlk, const double *uh, double *out)
{
    out[0] = ddlk;
    for (int j = 0; j < NV; j++)
        out[1 + j] = -dwk[j];
    out[17] = ip[IP_GS] * dwk[3];
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
    out[30] = ip[IP_GIM] * dwk[15];
    out[31] = dwk[13];
    out[32] = dwk[14];
    out[33] = dwk[0];
    out[34] = uh[0] * dwk[13] + uh[1] * dwk[14] + uh[2] * dwk[15];
    maskInactive(act, out);
}
// L' v (entries of inactive cones must be zero)
__device__ inline void LTmul(const double *ip, unsigned fm, const double *v, const double *uh, double *gw, double *gdl)
{
    *gdl = v[0];
    for (int j = 0; j < NV; j++)
        gw[j] = -v[1 + j];
    gw[3] += ip[IP_GS] * v[17];
    gw[1] += v[18];
    gw[2] += v[19];
    gw[8] += v[21];
    gw[9] += v[22];
    gw[11] += v[24];
    gw[12] += v[25];
    gw[13] += v[27] + v[31] + uh[0] * v[34];
    gw[14] += v[28] + v[32] + uh[1] * v[34];
    gw[15] += v[29] + ip[IP_GIM] * v[30] + uh[2] * v[34];
    gw[0] += v[33];
    for (int j = 0; j < NV; j++)
        if (fm & (1u << j))
            gw[j] = 0.;
}

__device__ inline double stageX(const double *wk, int j) { return j < 13 ? wk[j] : 0.; }
__device__ inline double stageU(const double *wk, int j) { return j < 3 ? wk[13 + j] : 0.; }

// dynamics residual of segment k: x_{k+1} - A x_k - B u_k - C u_{k+1} - S sigma - nu_k - Z_k
__device__ inline void dynRes(const Ctx &c, int k, const double *w0, const double *w1, const double *nuv, double sig,
                              double *out)
{
    const double *A = c.A + size_t(k) * NX * NX, *B = c.B + size_t(k) * NX * NU, *C = c.C + size_t(k) * NX * NU;
    for (int i = 0; i < NX; i++)
    {
        double acc = stageX(w1, i) - c.S[k * NX + i] * sig - nuv[i] - c.Z[k * NX + i];
        for (int j = 0; j < 13; j++)
            acc -= A[i * NX + j] * w0[j];
        for (int j = 0; j < 3; j++)
            acc -= B[i * NU + j] * w0[13 + j] + C[i * NU + j] * w1[13 + j];
        out[i] = acc;
    }
}
// entries of M_k = -[A|B] and N_k = [I|-C] in stage coordinates (fixed columns zeroed)
__device__ inline double Ment(const Ctx &c, int k, unsigned fm, int i, int j)
{
    if (fm & (1u << j))
        return 0.;
    return j < 13 ? -c.A[size_t(k) * NX * NX + i * NX + j] : -c.B[size_t(k) * NX * NU + i * NU + (j - 13)];
}
__device__ inline double Nent(const Ctx &c, int k, unsigned fmNext, int i, int j)
{
    if (fmNext & (1u << j))
        return 0.;
    return j < 13 ? (i == j ? 1. : 0.) : -c.C[size_t(k) * NX * NU + i * NU + (j - 13)];
}

// H += sum_ab c_a c_b W^-2_ab e_va e_vb'
__device__ inline void addConeH(double *H, double eta, const double *w, int d, const int *vars, const double *coef)
{
    const double e2 = 1. / (eta * eta);
    for (int a = 0; a < d; a++)
    {
        if (vars[a] < 0)
            continue;
        const double va = (a == 0) ? w[0] : -w[a];
        for (int b = 0; b < d; b++)
        {
            if (vars[b] < 0)
                continue;
            const double vb = (b == 0) ? w[0] : -w[b];
            double Wab = 2. * va * vb;
            if (a == b)
                Wab += (a == 0) ? -1. : 1.;
            H[vars[a] * NV + vars[b]] += coef[a] * coef[b] * Wab * e2;
        }
    }
}

// stage Hessian (delta_k eliminated), written row-major 16x16 to H (global); also hdd, hdw
__device__ inline void buildH(const Ctx &c, int k, bool identity, double *H)
{
    const unsigned fm = fixedMask(k, c.K), act = activeMask(k, c.K);
    double *st = c.st + size_t(k) * STREC;
    const double *eta = st + F_ETA, *wb = st + F_WB, *uh = st + F_UHAT;
    for (int i = 0; i < NV * NV; i++)
        H[i] = 0.;
    {
        const double e2 = 1. / (eta[0] * eta[0]);
        const double den = 2. * wb[0] * wb[0] - 1.;
        st[F_HDD] = den * e2;
        for (int j = 0; j < NV; j++)
            st[F_HDW + j] = (fm & (1u << j)) ? 0. : 2. * wb[0] * wb[1 + j] * e2;
        for (int i = 0; i < NV; i++)
            for (int j = 0; j < NV; j++)
                H[i * NV + j] = e2 * ((i == j ? 1. : 0.) - (2. / den) * wb[1 + i] * wb[1 + j]);
    }
    if (act & 2u)
    {
        const int v[3] = {3, 1, 2};
        const double cf[3] = {c.ip[IP_GS], 1., 1.};
        addConeH(H, eta[1], wb + C2, 3, v, cf);
    }
    if (act & 4u)
    {
        const int v[3] = {-1, 8, 9};
        const double cf[3] = {0., 1., 1.};
        addConeH(H, eta[2], wb + C3, 3, v, cf);
    }
    if (act & 8u)
    {
        const int v[3] = {-1, 11, 12};
        const double cf[3] = {0., 1., 1.};
        addConeH(H, eta[3], wb + C4, 3, v, cf);
    }
    if (act & 16u)
    {
        const int v[4] = {-1, 13, 14, 15};
        const double cf[4] = {0., 1., 1., 1.};
        addConeH(H, eta[4], wb + C5, 4, v, cf);
    }
    if (act & 32u)
    {
        const int v[3] = {15, 13, 14};
        const double cf[3] = {c.ip[IP_GIM], 1., 1.};
        addConeH(H, eta[5], wb + C6, 3, v, cf);
    }
    if (act & 64u)
        H[0] += identity ? 1. : st[F_Z + L1] / st[F_S + L1];
    if (act & 128u)
    {
        const double d = identity ? 1. : st[F_Z + L2] / st[F_S + L2];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                H[(13 + a) * NV + 13 + b] += d * uh[a] * uh[b];
    }
    for (int j = 0; j < NV; j++)
        if (fm & (1u << j))
        {
            for (int i = 0; i < NV; i++)
                H[i * NV + j] = H[j * NV + i] = 0.;
            H[j * NV + j] = 1.;
        }
}

// ---- cooperative tile helpers (LDS tiles, leading dimension 17) ----
constexpr int LD = 17;

// in-place lower Cholesky of the n x n tile in LDS; relative pivot floor 1e-14 (see structured_ipm.hpp)
__device__ inline void cholTile(double *Am, double *od, int n, int lane)
{
    if (lane < n)
        od[lane] = Am[lane * LD + lane];
    __syncthreads();
    for (int j = 0; j < n; j++)
    {
        if (lane == 0)
        {
            double d = Am[j * LD + j];
            const double orig = od[j];
            if (!(d > 1e-14 * orig))
                d = 1e-14 * orig;
            Am[j * LD + j] = sqrt(d);
        }
        __syncthreads();
        if (lane > j && lane < n)
            Am[lane * LD + j] /= Am[j * LD + j];
        __syncthreads();
        for (int e = lane; e < 256; e += WAVE)
        {
            const int i = e >> 4, cidx = e & 15;
            if (i < n && cidx > j && i >= cidx)
                Am[i * LD + cidx] -= Am[i * LD + j] * Am[cidx * LD + j];
        }
        __syncthreads();
    }
}

// D(16x16) += X' Y with X, Y 16x16 tiles in LDS (rows >= nrows treated as zero by the caller), on the
// FP64 matrix core: 4 x v_mfma_f64_16x16x4_f64.  Operand maps (cdna_hip_programming.md §3):
//   A-operand lane l: A[i=l&15][k=l>>4] ; B-operand: B[k=l>>4][j=l&15] ; D lane l reg r: (row (l>>4)+4r, col l&15)
__device__ inline d4_t mfmaXtY(const double *Xm, const double *Ym, int lane, d4_t acc)
{
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
    {
        const int kr = 4 * kk + hi; // contraction index = row of X and Y
        const double a = Xm[kr * LD + lo];
        const double b = Ym[kr * LD + lo];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
// D += X Y' : contraction over columns
__device__ inline d4_t mfmaXYt(const double *Xm, const double *Ym, int lane, d4_t acc)
{
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
    {
        const int kc = 4 * kk + hi;
        const double a = Xm[lo * LD + kc];
        const double b = Ym[lo * LD + kc];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// forward substitution with a lower-triangular n x n matrix whose ROW `lane` is in Lrow (lanes < n);
// g = right-hand side element of this lane; returns solution element (valid in lanes < n)
template <int n>
__device__ inline double trsvLower(const double *Lrow, double g, int lane)
{
    double x = 0.;
#pragma unroll
    for (int j = 0; j < n; j++)
    {
        const double xj_local = (lane == j) ? g / Lrow[j] : 0.;
        const double xj = __shfl(xj_local, j);
        if (lane == j)
            x = xj;
        if (lane > j && lane < n)
            g -= Lrow[j] * xj;
    }
    return x;
}
// backward substitution with L' where COLUMN `lane` of L is in Lcol (Lcol[q] = L[q][lane])
template <int n>
__device__ inline double trsvUpperT(const double *Lcol, double g, int lane)
{
    double x = 0.;
#pragma unroll
    for (int j = n - 1; j >= 0; j--)
    {
        const double xj_local = (lane == j) ? g / Lcol[j] : 0.;
        const double xj = __shfl(xj_local, j);
        if (lane == j)
            x = xj;
        if (lane < j)
            g -= Lcol[j] * xj; // L'[lane][j] = L[j][lane]
    }
    return x;
}

struct Shared
{
    double Pm[16 * LD]; // Phi / L
    double Ym[16 * LD];
    double Tm[16 * LD];
    double Zm[16 * LD];
    double Mm[16 * LD]; // M or N
    double od[16];
    double va[16];
    double vc[16];
};

// ---- block-tridiagonal factorisation (stage sweep) ----
// in: fac[k][0] = H_k (from buildH), seg EINV.  out: fac[k] = {L, Y, T, Z}.
__device__ inline void factorSweep(const Ctx &c, Shared &sh, int use_mfma)
{
    const int lane = c.lane, K = c.K;
    for (int k = 0; k < K; k++)
    {
        double *fk = c.fac + size_t(k) * FACREC;
        // Phi = H_k (+ Z_{k-1}' Z_{k-1})
        if (k > 0 && use_mfma)
        {
            d4_t acc = {0., 0., 0., 0.};
            acc = mfmaXtY(sh.Zm, sh.Zm, lane, acc);
            __syncthreads();
            const int col = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; r++)
            {
                const int row = (lane >> 4) + 4 * r;
                sh.Pm[row * LD + col] = fk[row * 16 + col] + acc[r];
            }
        }
        else
        {
            double tmp[4];
            for (int r = 0; r < 4; r++)
            {
                const int e = lane + WAVE * r, i = e >> 4, j = e & 15;
                double acc = fk[e];
                if (k > 0)
                    for (int q = 0; q < NL; q++)
                        acc += sh.Zm[q * LD + i] * sh.Zm[q * LD + j];
                tmp[r] = acc;
            }
            __syncthreads();
            for (int r = 0; r < 4; r++)
            {
                const int e = lane + WAVE * r, i = e >> 4, j = e & 15;
                sh.Pm[i * LD + j] = tmp[r];
            }
        }
        __syncthreads();
        cholTile(sh.Pm, sh.od, NV, lane);
        for (int r = 0; r < 4; r++)
        {
            const int e = lane + WAVE * r, i = e >> 4, j = e & 15;
            fk[e] = (j <= i) ? sh.Pm[i * LD + j] : 0.;
        }
        if (k == K - 1)
            break;
        // M tile (rows 14,15 zero)
        const unsigned fm = fixedMask(k, K), fmn = fixedMask(k + 1, K);
        for (int r = 0; r < 4; r++)
        {
            const int e = lane + WAVE * r, i = e >> 4, j = e & 15;
            sh.Mm[i * LD + j] = (i < NL) ? Ment(c, k, fm, i, j) : 0.;
        }
        __syncthreads();
        // Y = M L^-T : lane r < 14 does row r by forward substitution over columns
        if (lane < 16)
        {
            for (int j = 0; j < NV; j++)
            {
                double v = sh.Mm[lane * LD + j];
                for (int q = 0; q < j; q++)
                    v -= sh.Ym[lane * LD + q] * sh.Pm[j * LD + q];
                sh.Ym[lane * LD + j] = (lane < NL) ? v / sh.Pm[j * LD + j] : 0.;
            }
        }
        __syncthreads();
        // Theta = diag(Einv) + Y Y'   (rows/cols 14,15: identity padding)
        {
            const double *einv = c.sg + size_t(k) * SEGREC + G_EINV * NL;
            if (use_mfma)
            {
                d4_t acc = {0., 0., 0., 0.};
                acc = mfmaXYt(sh.Ym, sh.Ym, lane, acc);
                const int col = lane & 15;
#pragma unroll
                for (int r = 0; r < 4; r++)
                {
                    const int row = (lane >> 4) + 4 * r;
                    double v = acc[r];
                    if (row == col)
                        v += (row < NL) ? einv[row] : 1.;
                    sh.Tm[row * LD + col] = v;
                }
            }
            else
            {
                for (int r = 0; r < 4; r++)
                {
                    const int e = lane + WAVE * r, i = e >> 4, j = e & 15;
                    double acc = 0.;
                    for (int q = 0; q < NV; q++)
                        acc += sh.Ym[i * LD + q] * sh.Ym[j * LD + q];
                    if (i == j)
                        acc += (i < NL) ? einv[i] : 1.;
                    sh.Tm[i * LD + j] = acc;
                }
            }
        }
        __syncthreads();
        cholTile(sh.Tm, sh.od, NL, lane);
        // N tile
        for (int r = 0; r < 4; r++)
        {
            const int e = lane + WAVE * r, i = e >> 4, j = e & 15;
            sh.Mm[i * LD + j] = (i < NL) ? Nent(c, k, fmn, i, j) : 0.;
        }
        __syncthreads();
        // Z = T^-1 N : lane j < 16 does column j
        if (lane < 16)
        {
            for (int i = 0; i < NL; i++)
            {
                double v = sh.Mm[i * LD + lane];
                for (int q = 0; q < i; q++)
                    v -= sh.Tm[i * LD + q] * sh.Zm[q * LD + lane];
                sh.Zm[i * LD + lane] = v / sh.Tm[i * LD + i];
            }
            sh.Zm[14 * LD + lane] = 0.;
            sh.Zm[15 * LD + lane] = 0.;
        }
        __syncthreads();
        for (int r = 0; r < 4; r++)
        {
            const int e = lane + WAVE * r, i = e >> 4, j = e & 15;
            fk[256 + e] = sh.Ym[i * LD + j];
            fk[512 + e] = (i < NL && j < NL && j <= i) ? sh.Tm[i * LD + j] : (i == j ? 1. : 0.);
            fk[768 + e] = sh.Zm[i * LD + j];
        }
        __syncthreads();
    }
    __syncthreads();
}

// ---- block-tridiagonal solve: [dw; dlam] = T_mat^-1 [beta; rho] ----
// beta from stage field fBeta, rho from segment field gRho; result to stage field fOut, segment field gOut.
__device__ inline void blockSolve(const Ctx &c, Shared &sh, int fBeta, int gRho, int fOut, int gOut)
{
    const int lane = c.lane, K = c.K;
    double g = (lane < NV) ? c.st[F_BETA * 0 + fBeta + lane] : 0.; // stage 0
    for (int k = 0; k < K; k++)
    {
        const double *fk = c.fac + size_t(k) * FACREC;
        double row[16];
#pragma unroll
        for (int q = 0; q < NV; q++)
            row[q] = (lane < NV) ? fk[lane * 16 + q] : 1.;
        const double a = trsvLower<NV>(row, g, lane);
        if (lane < NV)
        {
            c.st[size_t(k) * STREC + F_AV + lane] = a;
            sh.va[lane] = a;
        }
        __syncthreads();
        if (k == K - 1)
            break;
        // gl = rho - Y a
        double gl = 0.;
        if (lane < NL)
        {
            gl = c.sg[size_t(k) * SEGREC + gRho * NL + lane];
            for (int q = 0; q < NV; q++)
                gl -= fk[256 + lane * 16 + q] * sh.va[q];
#pragma unroll
            for (int q = 0; q < NL; q++)
                row[q] = fk[512 + lane * 16 + q];
        }
        const double cc = trsvLower<NL>(row, gl, lane);
        if (lane < NL)
        {
            c.sg[size_t(k) * SEGREC + G_CV * NL + lane] = cc;
            sh.vc[lane] = cc;
        }
        __syncthreads();
        // g_next = beta_{k+1} + Z' c
        if (lane < NV)
        {
            double v = c.st[size_t(k + 1) * STREC + fBeta + lane];
            for (int r = 0; r < NL; r++)
                v += fk[768 + r * 16 + lane] * sh.vc[r];
            g = v;
        }
        __syncthreads();
    }
    // backward
    {
        const double *fk = c.fac + size_t(K - 1) * FACREC;
        double col[16];
#pragma unroll
        for (int q = 0; q < NV; q++)
            col[q] = (lane < NV) ? fk[q * 16 + lane] : 1.;
        const double a = (lane < NV) ? c.st[size_t(K - 1) * STREC + F_AV + lane] : 0.;
        const double x = trsvUpperT<NV>(col, a, lane);
        if (lane < NV)
        {
            c.st[size_t(K - 1) * STREC + fOut + lane] = x;
            sh.va[lane] = x;
        }
        __syncthreads();
    }
    for (int k = K - 2; k >= 0; k--)
    {
        const double *fk = c.fac + size_t(k) * FACREC;
        double col[16];
#pragma unroll
        for (int q = 0; q < 16; q++)
            col[q] = 1.;
        // t = Z x_{k+1} - c
        double t = 0.;
        if (lane < NL)
        {
            t = -c.sg[size_t(k) * SEGREC + G_CV * NL + lane];
            for (int j = 0; j < NV; j++)
                t += fk[768 + lane * 16 + j] * sh.va[j];
#pragma unroll
            for (int q = 0; q < NL; q++)
                col[q] = fk[512 + q * 16 + lane];
        }
        const double lk = trsvUpperT<NL>(col, t, lane);
        __syncthreads();
        if (lane < NL)
        {
            c.sg[size_t(k) * SEGREC + gOut * NL + lane] = lk;
            sh.vc[lane] = lk;
        }
        __syncthreads();
        double r = 0.;
        if (lane < NV)
        {
            r = c.st[size_t(k) * STREC + F_AV + lane];
            for (int i = 0; i < NL; i++)
                r -= fk[256 + i * 16 + lane] * sh.vc[i];
#pragma unroll
            for (int q = 0; q < NV; q++)
                col[q] = fk[q * 16 + lane];
        }
        const double x = trsvUpperT<NV>(col, r, lane);
        __syncthreads();
        if (lane < NV)
        {
            c.st[size_t(k) * STREC + fOut + lane] = x;
            sh.va[lane] = x;
        }
        __syncthreads();
    }
}

} // namespace ipm
} // namespace scpp
