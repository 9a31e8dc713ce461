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
    // SCvx mode (SCvxProblem.cpp:6-71) inside the same structure -- see oracle/structured_ipm.hpp (RQSocpInput::scvx):
    // delta_k is the constant trust_region, the state rows of the trust cone are zero padding, S = 0 decouples sigma
    IP_SCVX = 52,
    IP_TR = 53,
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
    F_HS = 522,   // [27] small Hessian blocks of the non-trust-region cones
    F_HC = 549,   // [2]  {1/eta1^2, 2/(2 w0^2 - 1)} of the trust-region cone
    F_WBK = 552,  // [17] W and delta of the last iterate that met the reduced tolerances (ECOS-style best iterate)
    STREC = 569
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
// per-stage factor record, PACKED (the kernel is HBM-throughput bound): Li lower triangle (136), Yt 16x14 (224),
// Ti lower triangle of the 14x14 block (105).  Z = Ti N is not stored: N = [I | -C] makes it 4 extra matrix-core
// instructions from Ti and the 42 entries of C.
constexpr int FAC_LI = 0, FAC_YT = 136, FAC_TI = 360, FACREC = 472;
constexpr int NRHS_MAX = 3;            // right-hand-side columns carried by one sweep
constexpr int SVREC = 2 * NRHS_MAX * 16; // saved forward intermediates (a, c) per stage

// Stage / segment records are stored FIELD-major: element f of stage k lives at st[f*64 + k], so that the
// lane == stage phases read and write fully coalesced 512-byte rows (one lane per stage).
constexpr int LANES = 64;
// row pitch (doubles) of the field-major records: one row = one field of all K stages.  K rounded up to even
// instead of 64 keeps rows 16-byte aligned and cuts the record traffic by 1 - K/64 (22 % at K = 50).
__host__ __device__ inline int recPitch(int K) { return (K + 1) & ~1; }
// field-major copy of the segment dynamics (A 14x14, B 14x4, C 14x4, s, z) for the lane = segment phases
constexpr int DY_A = 0, DY_B = NX * NX, DY_C = DY_B + NX * NU, DY_S = DY_C + NX * NU, DY_Z = DY_S + NX, DYNREC = DY_Z + NX; // 336
constexpr int GSAVE = 16; // wave-uniform scalars of the last solve (sigma, delta_sigma, n1 and their slacks / duals): warm start
__host__ __device__ inline size_t workspaceDoubles(int K)
{
    return size_t(recPitch(K)) * (STREC + SEGREC + DYNREC) + size_t(K) * (FACREC + SVREC) + GSAVE;
}

// Strided view of one lane's record, addressed through a buffer resource: every access is
//   buffer_load/store_dwordx2 v, v_lane_byte_offset, s[rsrc:rsrc+3], s_field_offset offen
// i.e. ONE 32-bit VGPR (lane*8) serves all fields and the field offsets live in SGPRs.  (Plain global
// pointers made the compiler keep ~70 loop-invariant 64-bit VGPR addresses -- one per 4 KB window of the
// field-major record -- and spill them.)
typedef unsigned int u32x2_t __attribute__((vector_size(8)));
struct SV
{
    __amdgpu_buffer_rsrc_t rsrc; // wave-uniform: the instance's stage or segment record block
    int lb;                      // lane offset in bytes
    int fo;                      // field offset (fields)
    int pb;                      // row pitch in bytes (wave-uniform)
    struct Ref
    {
        __amdgpu_buffer_rsrc_t rsrc;
        int lb, so;
        __device__ operator double() const
        {
            return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lb, so, 0));
        }
        __device__ const Ref &operator=(double x) const
        {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, x), rsrc, lb, so, 0);
            return *this;
        }
        __device__ const Ref &operator=(const Ref &o) const { return *this = double(o); }
        __device__ const Ref &operator+=(double x) const { return *this = double(*this) + x; }
        __device__ const Ref &operator-=(double x) const { return *this = double(*this) - x; }
        __device__ const Ref &operator*=(double x) const { return *this = double(*this) * x; }
    };
    __device__ Ref operator[](int i) const { return Ref{rsrc, lb, (fo + i) * pb}; }
    // lane-dependent field index: goes into the VGPR offset (a divergent SGPR offset would be resolved by a
    // waterfall loop over its distinct values)
    __device__ Ref dyn(int i) const { return Ref{rsrc, lb + (fo + i) * pb, 0}; }
    __device__ SV operator+(int o) const { return SV{rsrc, lb, fo + o, pb}; }
};
__device__ inline SV makeSV(double *block, int nfields, unsigned lane_index, int pitch)
{
    return SV{__builtin_amdgcn_make_buffer_rsrc(block, 0, nfields * pitch * 8, 0x00020000), int(lane_index * 8u), 0, pitch * 8};
}

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
    int pitch;   // row pitch of the field-major records (doubles)
    double *st;  // [STREC][64]   field-major
    double *sg;  // [SEGREC][64]  field-major
    double *dy;  // [DYNREC][64]  field-major copy of A,B,C,s,z
    double *fac; // [K][FACREC]
    double *sv;  // [K][SVREC]
    double *gsave; // [GSAVE]
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

template <class AV>
__device__ inline void maskInactive(unsigned act, AV v)
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
template <class AW, class AB, class AU, class AO>
__device__ inline void saff(const double *ip, unsigned act, AW wk, double dlk, AB wb, AU uh, AO out)
{
    const bool scvx = ip[IP_SCVX] != 0.;
    out[0] = scvx ? ip[IP_TR] : dlk;
    for (int j = 0; j < NV; j++)
        out[1 + j] = (scvx && j < 13) ? 0. : wb[j] - wk[j];
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
template <class AW, class AU, class AO>
__device__ inline void Lmul(const double *ip, unsigned act, AW dwk, double ddlk, AU uh, AO out)
{
    const bool scvx = ip[IP_SCVX] != 0.;
    out[0] = scvx ? 0. : ddlk;
    for (int j = 0; j < NV; j++)
        out[1 + j] = (scvx && j < 13) ? 0. : -dwk[j];
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
template <class AV, class AU>
__device__ inline void LTmul(const double *ip, unsigned fm, AV v, AU uh, double *gw, double *gdl)
{
    const bool scvx = ip[IP_SCVX] != 0.;
    *gdl = v[0];
    for (int j = 0; j < NV; j++)
        gw[j] = (scvx && j < 13) ? 0. : -v[1 + j];
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

template <class AW>
__device__ inline double stageX(AW wk, int j) { return j < 13 ? wk[j] : 0.; }

// dynamics residual of segment k: x_{k+1} - A x_k - B u_k - C u_{k+1} - S sigma - nu_k - Z_k
template <class A0, class A1, class AN>
__device__ inline void dynRes(const Ctx &c, int k, A0 w0, A1 w1, AN nuv, double sig, double *out)
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
// same on the field-major copy (coalesced, fully unrolled: all loads of a row are in flight together)
template <class A0, class A1, class AN>
__device__ inline void dynResF(const SV &dy, A0 w0, A1 w1, AN nuv, double sig, double *out)
{
    double x0[NV], u1[3];
#pragma unroll
    for (int j = 0; j < NV; j++)
        x0[j] = w0[j];
#pragma unroll
    for (int j = 0; j < 3; j++)
        u1[j] = w1[13 + j];
#pragma unroll
    for (int i = 0; i < NX; i++)
    {
        double acc = stageX(w1, i) - dy[DY_S + i] * sig - nuv[i] - dy[DY_Z + i];
#pragma unroll
        for (int j = 0; j < 13; j++)
            acc -= dy[DY_A + i * NX + j] * x0[j];
#pragma unroll
        for (int j = 0; j < 3; j++)
            acc -= dy[DY_B + i * NU + j] * x0[13 + j] + dy[DY_C + i * NU + j] * u1[j];
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
template <class AW>
__device__ inline void addConeH(double *H, double eta, AW w, int d, const int *vars, const double *coef)
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

// Per-stage Hessian data (delta_k eliminated) for the in-sweep tile build: F_HDD, F_HDW (delta elimination),
// F_HC = {1/eta1^2, 2/den} of the trust-region cone and F_HS = the small dense blocks contributed by the other
// cones / LP rows (layout: hsIndex in tile_engine.h).
__device__ inline int hsIndexK(int a, int b)
{
    if (a >= 1 && a <= 3 && b >= 1 && b <= 3)
        return (a - 1) * 3 + (b - 1);
    if (a >= 8 && a <= 9 && b >= 8 && b <= 9)
        return 9 + (a - 8) * 2 + (b - 8);
    if (a >= 11 && a <= 12 && b >= 11 && b <= 12)
        return 13 + (a - 11) * 2 + (b - 11);
    if (a >= 13 && b >= 13)
        return 17 + (a - 13) * 3 + (b - 13);
    if (a == 0 && b == 0)
        return 26;
    return -1;
}
template <class AW>
__device__ inline void addConeHs(double *Hs, double eta, AW w, int d, const int *vars, const double *coef)
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
            Hs[hsIndexK(vars[a], vars[b])] += coef[a] * coef[b] * Wab * e2;
        }
    }
}
__device__ inline void buildHs(const Ctx &c, int k, bool identity)
{
    const unsigned fm = fixedMask(k, c.K), act = activeMask(k, c.K);
    const SV st = makeSV(c.st, STREC, unsigned(k), c.pitch);
    const SV eta = st + F_ETA, wb = st + F_WB, uh = st + F_UHAT;
    double Hs[27];
    for (int i = 0; i < 27; i++)
        Hs[i] = 0.;
    {
        const bool scvx = c.ip[IP_SCVX] != 0.;
        const double e2 = 1. / (eta[0] * eta[0]);
        const double den = 2. * wb[0] * wb[0] - 1.;
        // SC: delta_k eliminated -> H = e2 (I - (2/den) w w').  SCvx: delta_k constant -> plain L'W^-2 L = e2 (I + 2 w w')
        // on the rows that exist (the mask is applied where the tile is built, sweeps.h buildHTile)
        st[F_HDD] = scvx ? 1. : den * e2;
        for (int j = 0; j < NV; j++)
            st[F_HDW + j] = ((fm & (1u << j)) || scvx) ? 0. : 2. * wb[0] * wb[1 + j] * e2;
        st[F_HC] = e2;
        st[F_HC + 1] = scvx ? -2. : 2. / den;
    }
    if (act & 2u)
    {
        const int v[3] = {3, 1, 2};
        const double cf[3] = {c.ip[IP_GS], 1., 1.};
        addConeHs(Hs, eta[1], wb + C2, 3, v, cf);
    }
    if (act & 4u)
    {
        const int v[3] = {-1, 8, 9};
        const double cf[3] = {0., 1., 1.};
        addConeHs(Hs, eta[2], wb + C3, 3, v, cf);
    }
    if (act & 8u)
    {
        const int v[3] = {-1, 11, 12};
        const double cf[3] = {0., 1., 1.};
        addConeHs(Hs, eta[3], wb + C4, 3, v, cf);
    }
    if (act & 16u)
    {
        const int v[4] = {-1, 13, 14, 15};
        const double cf[4] = {0., 1., 1., 1.};
        addConeHs(Hs, eta[4], wb + C5, 4, v, cf);
    }
    if (act & 32u)
    {
        const int v[3] = {15, 13, 14};
        const double cf[3] = {c.ip[IP_GIM], 1., 1.};
        addConeHs(Hs, eta[5], wb + C6, 3, v, cf);
    }
    if (act & 64u)
        Hs[26] += identity ? 1. : st[F_Z + L1] / st[F_S + L1];
    if (act & 128u)
    {
        const double d = identity ? 1. : st[F_Z + L2] / st[F_S + L2];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                Hs[17 + a * 3 + b] += d * uh[a] * uh[b];
    }
    for (int i = 0; i < 27; i++)
        st[F_HS + i] = Hs[i];
}

} // namespace ipm
} // namespace scpp
