// Main loop of the batched interior-point kernel (one wavefront per instance).
// Line-by-line device counterpart of oracle/structured_ipm.hpp::RQStructuredSocp::run (the scalar
// twin carries the derivation); see ipm_kernel.h for layout and tile helpers.
#pragma once
#include "ipm_kernel.h"
#include "sweeps.h"
#include <type_traits>

namespace scpp
{
namespace ipm
{

// Magnitude beyond which the primal cost or the complementarity gap marks an iterate as numerically broken (treated like a non-finite residual: the last
// iterate that met the reduced tolerances is returned, or the solve fails and a warm-started attempt is repeated cold).  Why it exists (round 5, instance
// 8392 of the bench's randomised states): one step of a warm-started solve came back with entries of 1e154 -- a broken factorisation, finite -- and
// because every residual is measured RELATIVE to the iterate's norm, the blown-up point showed pres = 0, passed the convergence test as "optimal" and
// handed inputs of 1e154 N to the next discretisation.  The sub-problems are nondimensional (costs 1e-2 .. 1e3; Rocket2D in SI units: 1e5, gaps up to
// 1e9 at the cold start): 1e30 is never a value of a working iterate.
#define IPM_BLOWN 1e30
// Step lengths of their own for the primal variables (x, s) and the dual ones (multipliers, z) -- round 6.  ECOS takes ONE step length, the largest
// that keeps s AND z in their cones; here s += alpha_p ds with the largest step that keeps s in the cone and z += alpha_d dz likewise (each with the
// same 0.99 / [1e-8, 0.999] safeguards), the way SDPT3-style solvers do.  The primal residuals still shrink by (1 - alpha_p (1 - sigma)), the dual ones by
// (1 - alpha_d (1 - sigma)); the centring parameter keeps ECOS's rule on the COMMON affine step length.  Measured on the scalar twin before it was
// built (16 RocketQuat K = 50 SCvx trajectories, tools/experiments/split_step_study.py): 366.1 -> 326.6 interior-point iterations per trajectory
// (16.0 -> 14.2 after an accepted step, 9.4 -> 8.9 after a rejection, 19.6 -> 17.5 for the first solve), all runs converged.  0: ECOS's common step
// (rounds 1 - 6a; the twin has the same switch, StructuredSettings::split_steps).
#ifndef IPM_SPLIT_STEPS
#define IPM_SPLIT_STEPS 1
#endif
#ifndef IPM_FUSE_UPDATE
#define IPM_FUSE_UPDATE 1 // the step x += alpha dx is applied by the NEXT residual pass on its way in (phResiduals<P, true>; round 6).  0: the two phases of
                          // rounds 1 - 5 (phUpdate, then phResiduals<P, false>) -- bitwise the same results (tests/tools/lib_equal.py on the GPU), 0.8 % slower
#endif
// A complementarity gap below -IPM_NEG_GAP is not a rounding artefact of an interior point (s and z inside the cone give s'z > 0): the iterate has left
// the cone.  Round 6: until then a negative gap counted as broken only once a fall-back iterate existed; without one, `gap < abstol`
// held trivially and a point with pres, dres below the tolerances and a gap of -1e29 returned status 0.  The magnitude test is two-sided now and a
// negative gap is broken with or without a fall-back, here, in the split schedule (ipm_split.h) and in the scalar twin (oracle/structured_ipm.hpp).
#define IPM_NEG_GAP 1e-6
__device__ inline bool ipmGapBroken(double gap) { return fabs(gap) > IPM_BLOWN || gap < -IPM_NEG_GAP; }

// wave-uniform scalars (every lane holds the same values)
struct Glob
{
    double sig, dsg, n1, sigbar;
    double ss, zs, s3, z3, sc3[3], zc3[3];
    double dsig, ddsg, dn1, dss, dzs, ds3, dz3, dsc3[3], dzc3[3];
    double seta, sw[3];       // sigma-cone scaling
    double hsig, Hsd, Hdd, schur;
    double lamC[3], dsC[3], dzC[3];
};

struct KernelArgs
{
    int B, K;
    double *X, *U, *sigma;       // [B][K][14], [B][K][4], [B]  (in: linearisation point, out: solution)
    const double *A, *Bm, *C, *S, *Z;
    const double *ip;            // [B][IP_N]
    const double *uhat;          // [B][K][3]
    double *ws;                  // [B][workspaceDoubles(K)]
    // SC bookkeeping (SCAlgorithm.cpp:100-131)
    double *wtrx;                // [B] current weight_trust_region_trajectory
    int *active;                 // [B] 1 while the SC loop of the instance is running
    int *converged, *sc_iters, *ipm_iters, *status;
    double *norm1_nu, *sum_delta, *delta_sigma;
    double nu_tol, delta_tol;
    int max_sc_iterations;
    int *warm;                   // [B] (may be null) 1: the workspace holds the primal-dual point of a successful previous solve
    int do_sc_update;            // 1: apply readSolution + convergence logic ; 0: plain sub-problem solve
    const int *dd_fresh;         // (may be null) [B] 0: A .. Z of the instance are what this workspace's PREVIOUS solve saw (SCvx: a re-solve after a rejected
                                 // candidate, SCvxAlgorithm.cpp:132-138) -- the field-major copy of dd and the data norm built on it are kept (round 6)
    double *Xold, *Uold;         // (may be null) SCvx: snapshot of the linearisation point taken before the solution
                                 // overwrites X / U (old_td = td, SCvxAlgorithm.cpp:77), [B][K][14] / [B][K][4]
    Settings opt;
    double *dbg;                 // optional [B][8]: pcost, gap, pres, dres, iters, status
};

struct Rhs
{
    double s, ds, n1, rhs3;
};

__device__ inline double dLP(bool identity, double sv, double zv) { return identity ? 1. : zv / sv; }

// per-lane preparation of the factorisation: segment scalars, stage Hessian, sigma block
template <class P>
__device__ inline void prepareFactor(const Ctx &c, bool identity, Glob &g)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    if (k < K - 1 && identity)
    {
        // initialisation (W = I): E^-1 = 1/2, q = 0.  In the main loop E^-1 is written by the predictor's right-hand-side phase
        // (rhsSegChunk<.., 0>, which holds the slacks anyway) and q is recomputed where it is used (segRhsRow)
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        const SV xs = makeSX(c.sx, L::XREC, K, unsigned(k));
#pragma unroll
        for (int i = 0; i < L::NL; i++)
        {
            xs[L::X_EINV + i] = 0.5;
            sg[G_QV * L::NL + i] = 0.;
        }
    }
    if (k < K)
        buildHs<P>(c, k, identity);
    {
        const double e2 = 1. / (g.seta * g.seta);
        const double vt[3] = {g.sw[0], -g.sw[1], -g.sw[2]};
        const double Ls[3] = {0., 0., 1.}, Ld[3] = {0.5, -0.5, 0.};
        double Hss = 0., Hsd = 0., Hdd = 0.;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
            {
                const double Wab = e2 * (2. * vt[a] * vt[b] + (a == b ? (a == 0 ? -1. : 1.) : 0.));
                Hss += Ls[a] * Wab * Ls[b];
                Hsd += Ls[a] * Wab * Ld[b];
                Hdd += Ld[a] * Wab * Ld[b];
            }
        Hss += dLP(identity, g.ss, g.zs);
        g.Hsd = Hsd;
        g.Hdd = Hdd;
        g.hsig = Hss - Hsd * Hsd / Hdd;
    }
}

// Reduced KKT solve, part 1: condensed right-hand side.  Input: L::F_BXW/L::F_BXD, G_BXNU/G_BXNUB/G_BY and b.
// Output: X_BETA (16) and X_RHO (NL) of the exchange records for the sweeps; returns the sigma-row rhs.
template <class P>
__device__ inline double kktPrep(const Ctx &c, bool identity, Glob &g, const Rhs &b)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    g.dz3 = -b.n1;
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        const SV xs = makeSX(c.sx, L::XREC, K, unsigned(k));
        for (int i = 0; i < L::NL; i++)
        {
            const double d1 = dLP(identity, sg[G_S1 * L::NL + i], sg[G_Z1 * L::NL + i]);
            const double d2 = dLP(identity, sg[G_S2 * L::NL + i], sg[G_Z2 * L::NL + i]);
            sg[G_DINV * L::NL + i] = 1. / (d1 + d2);
            const double bnb = sg[G_BXNUB * L::NL + i] - g.dz3;
            const double btn = sg[G_BXNU * L::NL + i] - sg[G_QV * L::NL + i] * bnb;
            sg[G_BNB * L::NL + i] = bnb;
            sg[G_BTN * L::NL + i] = btn;
            xs[L::X_RHO + i] = sg[G_BY * L::NL + i] + xs[L::X_EINV + i] * btn;
        }
    }
    if (k < K)
    {
        const SV st = makeSV(c.st, L::STREC, unsigned(k), c.pitch);
        const unsigned fm = L::fixedMask(k, K);
        const SV xs = makeSX(c.sx, L::XREC, K, unsigned(k));
        for (int j = 0; j < NV; j++)
            xs[L::X_BETA + j] = (fm & (1u << j)) ? 0. : st[L::F_BXW + j] - st[L::F_HDW + j] * st[L::F_BXD] / st[L::F_HDD];
    }
    return b.s - g.Hsd * b.ds / g.Hdd;
}

// border column products: schur complement of sigma (after the border column has been solved)
template <class P>
__device__ inline void borderSchur(const Ctx &c, Glob &g)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    double acc = 0.;
    if (k < K - 1)
        for (int i = 0; i < L::NL; i++)
            acc += -c.S[k * P::NX + i] * c.sx[size_t(k) * L::XREC + L::X_BCL + i];
    acc = wave_sum(acc);
    g.schur = g.hsig - acc;
}

// part 2: after the sweeps left T_mat^-1 [beta;rho] in X_VW / X_VL: border correction and recovery of the
// eliminated variables.
template <class P>
__device__ inline void kktFinish(const Ctx &c, bool identity, Glob &g, const Rhs &b, double bts)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    double cv = 0.;
    if (k < K - 1)
        for (int i = 0; i < L::NL; i++)
            cv += -c.S[k * P::NX + i] * c.sx[size_t(k) * L::XREC + L::X_VL + i];
    cv = wave_sum(cv);
    g.dsig = (bts - cv) / g.schur;
    g.ddsg = (b.ds - g.Hsd * g.dsig) / g.Hdd;
    double sumnb = 0.;
    if (k < K)
    {
        const SV st = makeSV(c.st, L::STREC, unsigned(k), c.pitch);
        const SV xs = makeSX(c.sx, L::XREC, K, unsigned(k));
        double acc = 0.;
        for (int j = 0; j < NV; j++)
        {
            const double d = xs[L::X_VW + j] - xs[L::X_BCW + j] * g.dsig;
            st[L::F_DW + j] = d;
            acc += st[L::F_HDW + j] * d;
        }
        st[L::F_DDL] = (c.ip[IP_SCVX] != 0.) ? 0. : (st[L::F_BXD] - acc) / st[L::F_HDD];
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        const SV xs = makeSX(c.sx, L::XREC, K, unsigned(k));
        for (int i = 0; i < L::NL; i++)
        {
            const double dl = xs[L::X_VL + i] - xs[L::X_BCL + i] * g.dsig;
            sg[G_DLAM * L::NL + i] = dl;
            const double dnu = xs[L::X_EINV + i] * (dl + sg[G_BTN * L::NL + i]);
            const double dnub = sg[G_BNB * L::NL + i] * sg[G_DINV * L::NL + i] - sg[G_QV * L::NL + i] * dnu;
            sg[G_DNU * L::NL + i] = dnu;
            sg[G_DNUB * L::NL + i] = dnub;
            sumnb += dnub;
        }
    }
    sumnb = wave_sum(sumnb);
    const double w3sq = identity ? 1. : g.s3 / g.z3;
    g.dn1 = sumnb - w3sq * g.dz3 - b.rhs3;
}

__device__ inline RhsSpec specBorderPlus()
{
    RhsSpec sp;
    sp.n = 2;
    return sp;
}
__device__ inline RhsSpec specSingle()
{
    RhsSpec sp;
    sp.n = 1;
    return sp;
}

template <class P>
__device__ inline void applyPrimalStep(const Ctx &c, Glob &g, double alpha)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    if (k < K)
    {
        const SV st = makeSV(c.st, L::STREC, unsigned(k), c.pitch);
        const unsigned fm = L::fixedMask(k, K);
#pragma unroll
        for (int j = 0; j < NV; j++)
        {
            const double d = st[L::F_DW + j];
            st[L::F_W + j] += (fm & (1u << j)) ? 0. : alpha * d;
        }
        st[L::F_DL] += alpha * st[L::F_DDL];
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        for (int i = 0; i < L::NL; i++)
        {
            sg[G_NU * L::NL + i] += alpha * sg[G_DNU * L::NL + i];
            sg[G_NUB * L::NL + i] += alpha * sg[G_DNUB * L::NL + i];
        }
    }
    g.sig += alpha * g.dsig;
    g.dsg += alpha * g.ddsg;
    g.n1 += alpha * g.dn1;
}

// affine slacks of everything at the current primal point -> stage field fOut, segment fields g1,g2; scalars
template <class P>
__device__ inline void evalAllSaff(const Ctx &c, const Glob &g, int fOut, int g1, int g2, double &os, double &o3, double *oc)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    double sumnb = 0.;
    if (k < K)
    {
        const SV st = makeSV(c.st, L::STREC, unsigned(k), c.pitch);
        saff<P>(c.ip, L::activeMask(k, K), st + L::F_W, st[L::F_DL], st + L::F_WBAR, st + L::F_UHAT, st + fOut);
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        for (int i = 0; i < L::NL; i++)
        {
            const double nu = sg[G_NU * L::NL + i], nub = sg[G_NUB * L::NL + i];
            sg[g1 * L::NL + i] = nub - nu;
            sg[g2 * L::NL + i] = nub + nu;
            sumnb += nub;
        }
    }
    sumnb = wave_sum(sumnb);
    os = g.sig - 0.001;
    o3 = g.n1 - sumnb;
    oc[0] = 0.5 + 0.5 * g.dsg;
    oc[1] = 0.5 - 0.5 * g.dsg;
    oc[2] = g.sig - g.sigbar;
}

// ECOS bring2cone over the whole product cone: stage field f, segment fields g1,g2, scalars
template <class P>
__device__ inline void bring2cone(const Ctx &c, double gamma, int f, int g1, int g2, double &vs, double &v3, double *vc)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    double alpha = -gamma;
    if (k < K)
    {
        const unsigned act = L::activeMask(k, K);
        const SV v = makeSV(c.st, L::STREC, unsigned(k), c.pitch) + f;
#pragma unroll
        for (int cix = 0; cix < L::NCONES; cix++)
            if (act & (1u << cix))
            {
                const SV r = v + L::CONE_OFF.v[cix];
                double nrm = 0.;
                for (int i = 1; i < L::CONE_DIM.v[cix]; i++)
                    nrm += r[i] * r[i];
                const double cres = r[0] - sqrt(nrm);
                if (cres <= 0. && -cres > alpha)
                    alpha = -cres;
            }
#pragma unroll
        for (int l = 0; l < P::NLP; l++)
            if ((act & (1u << (L::NCONES + l))) && v[L::LP0 + l] <= 0. && -v[L::LP0 + l] > alpha)
                alpha = -v[L::LP0 + l];
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        for (int i = 0; i < L::NL; i++)
        {
            const double a = sg[g1 * L::NL + i], b = sg[g2 * L::NL + i];
            if (a <= 0. && -a > alpha)
                alpha = -a;
            if (b <= 0. && -b > alpha)
                alpha = -b;
        }
    }
    if (vs <= 0. && -vs > alpha)
        alpha = -vs;
    if (v3 <= 0. && -v3 > alpha)
        alpha = -v3;
    {
        const double cres = vc[0] - sqrt(vc[1] * vc[1] + vc[2] * vc[2]);
        if (cres <= 0. && -cres > alpha)
            alpha = -cres;
    }
    alpha = wave_max(alpha) + 1.;
    if (k < K)
    {
        const unsigned act = L::activeMask(k, K);
        const SV v = makeSV(c.st, L::STREC, unsigned(k), c.pitch) + f;
#pragma unroll
        for (int cix = 0; cix < L::NCONES; cix++)
            if (act & (1u << cix))
                v[L::CONE_OFF.v[cix]] += alpha;
#pragma unroll
        for (int l = 0; l < P::NLP; l++)
            if (act & (1u << (L::NCONES + l)))
                v[L::LP0 + l] += alpha;
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        for (int i = 0; i < L::NL; i++)
        {
            sg[g1 * L::NL + i] += alpha;
            sg[g2 * L::NL + i] += alpha;
        }
    }
    vs += alpha;
    v3 += alpha;
    vc[0] += alpha;
}


// warm start: v += (theta + largest violation) e over the whole product cone (oracle/structured_ipm.hpp: shiftToCone)
template <class P>
__device__ inline void shiftToCone(const Ctx &c, double theta, int f, int g1, int g2, double &vs, double &v3, double *vc)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    double alpha = 0.;
    if (k < K)
    {
        const unsigned act = L::activeMask(k, K);
        const SV v = makeSV(c.st, L::STREC, unsigned(k), c.pitch) + f;
#pragma unroll
        for (int cix = 0; cix < L::NCONES; cix++)
            if (act & (1u << cix))
            {
                const SV r = v + L::CONE_OFF.v[cix];
                double nrm = 0.;
                for (int i = 1; i < L::CONE_DIM.v[cix]; i++)
                    nrm += r[i] * r[i];
                const double cres = r[0] - sqrt(nrm);
                if (-cres > alpha)
                    alpha = -cres;
            }
#pragma unroll
        for (int l = 0; l < P::NLP; l++)
            if ((act & (1u << (L::NCONES + l))) && -v[L::LP0 + l] > alpha)
                alpha = -v[L::LP0 + l];
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        for (int i = 0; i < L::NL; i++)
        {
            const double a = sg[g1 * L::NL + i], b = sg[g2 * L::NL + i];
            if (-a > alpha)
                alpha = -a;
            if (-b > alpha)
                alpha = -b;
        }
    }
    if (-vs > alpha)
        alpha = -vs;
    if (-v3 > alpha)
        alpha = -v3;
    {
        const double cres = vc[0] - sqrt(vc[1] * vc[1] + vc[2] * vc[2]);
        if (-cres > alpha)
            alpha = -cres;
    }
    alpha = wave_max(alpha) + theta;
    if (k < K)
    {
        const unsigned act = L::activeMask(k, K);
        const SV v = makeSV(c.st, L::STREC, unsigned(k), c.pitch) + f;
#pragma unroll
        for (int cix = 0; cix < L::NCONES; cix++)
            if (act & (1u << cix))
                v[L::CONE_OFF.v[cix]] += alpha;
#pragma unroll
        for (int l = 0; l < P::NLP; l++)
            if (act & (1u << (L::NCONES + l)))
                v[L::LP0 + l] += alpha;
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        for (int i = 0; i < L::NL; i++)
        {
            sg[g1 * L::NL + i] += alpha;
            sg[g2 * L::NL + i] += alpha;
        }
    }
    vs += alpha;
    v3 += alpha;
    vc[0] += alpha;
}

// f(integral_constant<int, C>) for every cone C of the model's problem (0 = trust region, then the table's cones)
template <class P, int C = 0, class F>
__device__ inline void forEachCone(F &&f)
{
    if constexpr (C < Lay<P>::NCONES)
    {
        f(std::integral_constant<int, C>{});
        forEachCone<P, C + 1>(f);
    }
}

// ---- per-cone work on register arrays (compile-time offset / dimension) ----
// segment rows [0, NL) in balanced chunks of at most CH rows: f(integral_constant<I0>, integral_constant<N>) per chunk
template <class P, int CH, int J = 0, class F>
__device__ inline void forSegChunks(F &&f)
{
    constexpr int NL = Lay<P>::NL, NCH = (NL + CH - 1) / CH;
    if constexpr (J < NCH)
    {
        constexpr int base = NL / NCH, rem = NL % NCH;
        constexpr int I0 = J * base + (J < rem ? J : rem), N = base + (J < rem ? 1 : 0);
        f(std::integral_constant<int, I0>{}, std::integral_constant<int, N>{});
        forSegChunks<P, CH, J + 1>(f);
    }
}
#ifndef IPM_DIR_CHUNK
#define IPM_DIR_CHUNK 7 // (5 until round 4: with the recomputing chunks 7 rows = two chunks measured +0.4 %)
#endif
#ifndef IPM_RHS_CHUNK
#define IPM_RHS_CHUNK 7 // (5 until round 4: with the recomputing chunks 7 rows = two chunks measured +0.4 %)
#endif
#ifndef IPM_RES_CHUNK
#define IPM_RES_CHUNK 7
#endif
#ifndef IPM_UPD_CHUNK
#define IPM_UPD_CHUNK 7 // (5 until round 4: with the recomputing chunks 7 rows = two chunks measured +0.4 %)
#endif
// chunk size of a kernel variant: the workspace-resident variants run three wavefronts per SIMD (168 VGPRs) and take shorter chunks
#ifndef IPM_DIR_CHUNK_W
#define IPM_DIR_CHUNK_W 5
#endif
#ifndef IPM_RHS_CHUNK_W
#define IPM_RHS_CHUNK_W 5
#endif
#ifndef IPM_RES_CHUNK_W
#define IPM_RES_CHUNK_W 5
#endif
#ifndef IPM_UPD_CHUNK_W
#define IPM_UPD_CHUNK_W 5
#endif
template <class P>
constexpr int chunkFor(int resident, int split)
{
    return SegInLds<P>::value ? resident : split;
}
// Rows 1 .. NP of a vector that starts at the trust-region cone are STRUCTURAL ZEROS in SCvx mode (the state rows of the cone:
// saff / Lmul produce 0 there, and every cone operation maps zero rows to zero rows).  They are accessed through a second view
// `rz` of the same record whose lane offset lies beyond the record block in SCvx mode: the buffer hardware then returns 0 for
// the load and drops the store -- 13 of the 33 slack rows of a RocketQuat stage cost no memory traffic, without a branch.
// (In SC mode rz == r.)  NP = 0: a plain vector.
__device__ inline SV padView(const SV &r, bool scvx)
{
    return SV{r.rsrc, scvx ? VO_OOB : r.lb, r.fo, r.pb};
}
template <int N, int NP>
__device__ inline void ldPad(const SV &r, const SV &rz, int f, double (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++)
        v[i] = (i >= 1 && i <= NP) ? double(rz[f + i]) : double(r[f + i]);
}
// entries I0 .. I0 + CN - 1 of a padded stage vector (ldPad's rule per entry) into an array of their own
template <int NP, int I0, int CN>
__device__ inline void ldPadPart(const SV &r, const SV &rz, int f, double (&v)[CN])
{
#pragma unroll
    for (int i = 0; i < CN; i++)
        v[i] = (I0 + i >= 1 && I0 + i <= NP) ? double(rz[f + I0 + i]) : double(r[f + I0 + i]);
}
template <int N, int NP>
__device__ inline void stPad(const SV &r, const SV &rz, int f, const double (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        if (i >= 1 && i <= NP)
            rz[f + i] = v[i];
        else
            r[f + i] = v[i];
    }
}
__device__ inline bool scvxMode(const double *ip) { return uniformInt(int(ip[IP_SCVX] != 0.)) != 0; }
// one cone's part of a stage vector: cone 0 (OFF == 0) is the trust-region cone with its NXV state rows
template <class P, int OFF, int D>
__device__ inline void ldv(const SV &st, int f, double (&v)[D], const SV &stz)
{
    ldPad<D, (OFF == 0 ? P::NXV : 0)>(st, stz, f + OFF, v);
}
template <class P, int OFF, int D>
__device__ inline void stv(const SV &st, int f, const double (&v)[D], const SV &stz)
{
    stPad<D, (OFF == 0 ? P::NXV : 0)>(st, stz, f + OFF, v);
}
// NT scaling of one cone + lambda = W z ; returns 1 if the iterate left the cone
template <class P, int OFF, int D>
__device__ inline int coneScaling(const SV &st, int cix, const SV &stz)
{
    using L = Lay<P>;
    double s[D], z[D], w[D], ls[D], eta;
    ldv<P, OFF, D>(st, L::F_S, s, stz);
    ldv<P, OFF, D>(st, L::F_Z, z, stz);
    LOADS_ISSUED();
    if (!cone::nt_scalingS<D>(s, z, eta, w))
        return 1;
    cone::applyWS<D>(eta, w, z, ls);
    st[L::F_ETA + cix] = eta;
    stv<P, OFF, D>(st, L::F_WB, w, stz);
    stv<P, OFF, D>(st, L::F_LS, ls, stz);
    return 0;
}
// t = W^-2 rz' + W^-1(lambda \ ds) of one cone
template <class P, int OFF, int D>
__device__ inline void coneT(const SV &st, int cix, int pass, double om, double sigmu, double *tout, const SV &stz)
{
    using L = Lay<P>;
    double w[D], rz[D], b2[D], t[D];
    // one load group: what both passes read, then what this pass reads (z | the scaled affine directions and lambda)
    double q1[D], q3[D];
    const double eta = st[L::F_ETA + cix];
    ldv<P, OFF, D>(st, L::F_WB, w, stz);
    ldv<P, OFF, D>(st, L::F_RZ, rz, stz);
    if (pass == 0)
        ldv<P, OFF, D>(st, L::F_Z, q1, stz);
    else
    {
        ldv<P, OFF, D>(st, L::F_DSS, q1, stz); // (W^-1 ds_aff) o (W dz_aff), formed by the predictor's direction phase
        ldv<P, OFF, D>(st, L::F_LS, q3, stz);
    }
    LOADS_ISSUED();
#pragma unroll
    for (int i = 0; i < D; i++)
        rz[i] *= om;
    cone::applyWinv2S<D>(eta, w, rz, b2);
    if (pass == 0)
    {
        const double(&z)[D] = q1;
#pragma unroll
        for (int i = 0; i < D; i++)
            t[i] = b2[i] - z[i];
    }
    else
    {
        const double(&ls)[D] = q3;
        double dsv[D], aa[D];
#pragma unroll
        for (int i = 0; i < D; i++)
            dsv[i] = -q1[i];
        dsv[0] += sigmu;
        cone::conicDivisionS<D>(ls, dsv);
#pragma unroll
        for (int i = 0; i < D; i++)
            dsv[i] -= ls[i];
        cone::applyWinvS<D>(eta, w, dsv, aa);
#pragma unroll
        for (int i = 0; i < D; i++)
            t[i] = b2[i] + aa[i];
    }
    stv<P, OFF, D>(st, L::F_TZ, t, stz);
#pragma unroll
    for (int i = 0; i < D; i++)
        tout[OFF + i] = t[i]; // stays in registers for the L't product of the same phase (no re-read of F_TZ)
}
// dz = -W^-2 L dx + t ; ds = -rz' + L dx ; scaled directions ; returns 1/alpha_max of this cone's slack direction (the multiplier direction's in ainv_dual)
// store_final = false (predictor pass): only the scaled directions are needed afterwards (corrector term)
template <class P, int OFF, int D>
__device__ inline double coneDir(const SV &st, int cix, double om, const double *Ldall, bool store_final, const SV &stz, double &ainv_dual)
{
    using L = Lay<P>;
    double w[D], Ld[D], aa[D], t[D], rz[D], dz[D], ds[D], dss[D], dzs[D], ls[D];
    const double eta = st[L::F_ETA + cix];
    ldv<P, OFF, D>(st, L::F_WB, w, stz);
    ldv<P, OFF, D>(st, L::F_TZ, t, stz);
    ldv<P, OFF, D>(st, L::F_RZ, rz, stz);
    ldv<P, OFF, D>(st, L::F_LS, ls, stz);
    LOADS_ISSUED();
#pragma unroll
    for (int i = 0; i < D; i++)
        Ld[i] = Ldall[OFF + i];
    cone::applyWinv2S<D>(eta, w, Ld, aa);
#pragma unroll
    for (int i = 0; i < D; i++)
    {
        dz[i] = -aa[i] + t[i];
        ds[i] = -om * rz[i] + Ld[i];
    }
    cone::applyWinvS<D>(eta, w, ds, dss);
    cone::applyWS<D>(eta, w, dz, dzs);
    if (store_final)
    {
        stv<P, OFF, D>(st, L::F_DZ, dz, stz);
        stv<P, OFF, D>(st, L::F_DS, ds, stz);
    }
    else
    {
        // the scaled affine directions feed the corrector's right-hand side (coneT, pass 1) through their conic product only
        double prod[D];
        cone::conicProductS<D>(dss, dzs, prod);
        stv<P, OFF, D>(st, L::F_DSS, prod, stz);
    }
    // 1 / alpha_max of the slack direction (returned) and of the multiplier direction (running maximum of the caller)
    const double a1 = cone::stepInvS<D>(ls, dss), a2 = cone::stepInvS<D>(ls, dzs);
    ainv_dual = a2 > ainv_dual ? a2 : ainv_dual;
    return a1;
}
template <class P, int OFF, int D>
__device__ inline void zeroT(const SV &st)
{
    using L = Lay<P>;
#pragma unroll
    for (int i = 0; i < D; i++)
        st[L::F_TZ + OFF + i] = 0.;
}


// =====================================================================================================
// The solver is split into OUT-OF-LINE phases.  Each phase gets its own register allocation (the monolithic
// kernel kept >1000 values alive and spilled inside every hot loop); what survives a phase boundary lives in
// the records in global memory or in the two small wave-uniform structs Glob / Iter, which the kernel keeps in LDS (one
// copy per wavefront, 1.4 KB).  The per-iteration phases read their fields where they need them and write back only what
// they change (PUT); the once-per-solve phases copy them in and out whole (loadPriv / storePriv).
// PHASE_FN: static + noinline + disable_tail_calls.  LLVM skips the save / restore of callee-saved VGPRs for functions with
// internal linkage that are never the target of a tail call, and every call that passes no pointer into the caller's stack
// frame is a tail-call candidate unless the attribute says otherwise.  Without it each heavy phase saved and restored 108
// VGPRs (57 KB of scratch traffic per call and wavefront) -- DESIGN.md 4.2 / 5.0.
// =====================================================================================================
#define PRIV LDSP
#ifdef SCPP_HIP_EMU
#define PHASE_FN inline
#else
#define PHASE_FN static __device__ __attribute__((noinline, disable_tail_calls))
#endif

// wave-uniform state of one solve
struct Iter
{
    double wtrx, w_t, w_trt, w_vc, sigbar, gamma;
    double resx0, resy0, resz0;
    double mu, gap, pres, dres, pcost;
    double rzs, rz3, rzc[3], rxs, rxds, rxn1;
    double sigma_c, alpha, alpha_d, tzs, tzc[3]; // alpha: step length of the primal variables, alpha_d: of the dual ones (== alpha without IPM_SPLIT_STEPS)
    Rhs b;
    double bts;
    // ECOS-style safeguarding: scalars of the last iterate that met the reduced tolerances (its W / delta are in L::F_WBK)
    double bk_sig, bk_dsg, bk_n1, pres_prev;
    double part_ainv, part_ainv_d, part_fin; // partial step-length bounds (primal, dual) / finiteness check handed from phDirStage to phDirSeg
    int D, bad, bk_valid;
    int common_step; // != 0: this attempt takes ECOS's common step length (the repeat of an attempt that failed with split ones, ipmSolveInstance)
};

template <class T>
__device__ inline T loadPriv(const PRIV T *p)
{
    T t;
    __builtin_memcpy(&t, p, sizeof(T));
    return t;
}
// a second, late copy-in: the barrier keeps the compiler from merging it with loads at the top of the phase, so wave-uniform
// values that are only needed in a phase's tail do not occupy VGPRs across its load / compute / store groups
template <class T>
__device__ inline T reloadPriv(const PRIV T *p)
{
#ifndef SCPP_HIP_EMU
    asm volatile("" ::: "memory");
#endif
    return loadPriv(p);
}
template <class T>
__device__ inline void storePriv(PRIV T *p, const T &t)
{
    WAVE_SYNC(); // every lane has taken its copy before the (wave-uniform) new value goes in
    if (threadIdx.x == 0)
        __builtin_memcpy(p, &t, sizeof(T));
}

// Field-wise write-back: a phase that copies the whole struct out again keeps every field alive in VGPRs from its first
// instruction to its last (~80 wave-uniform doubles = 160 VGPRs in the per-iteration phases, measured in the ISA: all ds_read
// at the top, all ds_write at the bottom), which is what pushed those phases into the callee-saved registers and into scratch.
// The per-iteration phases therefore store only what they change: PUT(ptr, obj, field) inside PUT_BEGIN / PUT_END.
template <class T>
__device__ inline void putv(PRIV T &dst, const T &src)
{
    dst = src;
}
template <class T, int N>
__device__ inline void putv(PRIV T (&dst)[N], const T (&src)[N])
{
    for (int i = 0; i < N; i++)
        dst[i] = src[i];
}
#define PUT(ptr, obj, f) putv((ptr)->f, (obj).f)
#define PUT_BEGIN()                                                                                                    \
    WAVE_SYNC(); /* every lane has read the old values */                                                              \
    if (threadIdx.x == 0)                                                                                              \
    {
#define PUT_END()                                                                                                      \
    }                                                                                                                  \
    WAVE_SYNC()

// lane-local views used by every phase
struct Views
{
    int k, K;
    bool vst, vsg;
    unsigned fm, act;
    SV st, stN, sg, sgP, dy, dyP;
    SV xs; // this lane's exchange record (stage-major)
};
template <class P>
__device__ inline Views makeViews(const Ctx &c)
{
    using L = Lay<P>;
    const int k = c.lane, K = c.K;
    Views v{k,
            K,
            k < K,
            k < K - 1,
            (k < K) ? L::fixedMask(k, K) : 0u,
            (k < K) ? L::activeMask(k, K) : 0u,
            makeSV(c.st, L::STREC, unsigned(k < K ? k : 0), c.pitch),
            makeSV(c.st, L::STREC, unsigned(k < K - 1 ? k + 1 : 0), c.pitch),
            makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k < K - 1 ? k : 0), c.pitch),
            makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k > 0 && k < K ? k - 1 : 0), c.pitch),
            makeSV(c.dy, L::DYNREC, unsigned(k < K - 1 ? k : 0), c.pitch),
            makeSV(c.dy, L::DYNREC, unsigned(k > 0 && k < K ? k - 1 : 0), c.pitch),
            makeSX(c.sx, L::XREC, K, unsigned(k < K ? k : 0))};
    return v;
}

// ---- setup: clear records, field-major copy of the dynamics, trust-region centre, fixed values ----
template <class P>
PHASE_FN void phSetup(const PRIV Ctx *cin, const double *Xin, const double *Uin, const double *uhatIn, PRIV Glob *gp, PRIV Iter *ip_, int warmIn, int ddSameIn)
{
    EMU_PHASE("phSetup");
    using L = Lay<P>;
    constexpr int NX = P::NX, NU = P::NU;
    const bool warm = uniformInt(warmIn) != 0;
    const bool dd_same = uniformInt(ddSameIn) != 0; // the workspace's field-major copy of A .. Z is still that of these data (167 KB not rewritten per re-solve)
    const Ctx c = uniformCtx(cin);
    const double *X = uniformPtr(Xin), *U = uniformPtr(Uin), *uhat = uniformPtr(uhatIn);
    const Views v = makeViews<P>(c);
    const int k = v.k, K = v.K;
    const SV &st = v.st, &sg = v.sg, &dy = v.dy;
    const double *ip = c.ip;
    Glob g;
    Iter it = loadPriv(ip_);
    g.sig = g.dsg = g.n1 = 0.;
    g.sigbar = it.sigbar;
    g.ss = g.zs = g.s3 = g.z3 = 1.;
    for (int i = 0; i < 3; i++)
        g.sc3[i] = g.zc3[i] = g.dsc3[i] = g.dzc3[i] = g.lamC[i] = g.dsC[i] = g.dzC[i] = 0.;
    g.dsig = g.ddsg = g.dn1 = g.dss = g.dzs = g.ds3 = g.dz3 = 0.;
    g.hsig = g.Hsd = g.Hdd = g.schur = 0.;
    g.seta = 1.;
    g.sw[0] = 1.;
    g.sw[1] = g.sw[2] = 0.;
    if (warm)
    {
        // warm start: primal point, slacks and duals of the previous solve stay in the records; the wave-uniform part
        // was saved at the end of that solve
        const double *gs = c.gsave;
        g.sig = gs[0];
        g.dsg = gs[1];
        g.n1 = gs[2];
        g.ss = gs[3];
        g.zs = gs[4];
        g.s3 = gs[5];
        g.z3 = gs[6];
        for (int i = 0; i < 3; i++)
        {
            g.sc3[i] = gs[7 + i];
            g.zc3[i] = gs[10 + i];
        }
    }
    int Dcount = 0;
    // device memory is not zero-initialised: clear this lane's records (entries of inactive cones are
    // never written afterwards but are swept by the vector updates)
    if (v.vst && !warm)
    {
        for (int i = 0; i < L::STREC; i++)
            st[i] = 0.;
        for (int i = 0; i < L::XREC; i++)
            v.xs[i] = 0.;
    }
    if (v.vsg)
    {
        if (!warm)
            for (int i = 0; i < (G_NFIELDS * L::NL); i++)
                sg[i] = 0.;
        for (int i = 0; i < L::NL; i++)
            v.xs[L::X_S + i] = c.S[size_t(k) * NX + i];
        if (!dd_same)
        {
            // field-major copy of this segment's dynamics (read once row-major, re-read coalesced every iteration)
            EMU_TRAFFIC_MANUAL("dd A, B, C, S, Z row-major (plain pointers)", NX * NX + 2 * NX * NU + 3 * NX, false);
            const double *Ak = c.A + size_t(k) * NX * NX, *Bk = c.B + size_t(k) * NX * NU, *Ck = c.C + size_t(k) * NX * NU;
            for (int e = 0; e < NX * NX; e++)
                dy[L::DY_A + e] = Ak[e];
            for (int e = 0; e < NX * NU; e++)
            {
                dy[L::DY_B + e] = Bk[e];
                dy[L::DY_C + e] = Ck[e];
            }
            for (int e = 0; e < NX; e++)
            {
                dy[L::DY_S + e] = c.S[k * NX + e];
                dy[L::DY_Z + e] = c.Z[k * NX + e];
            }
        }
    }
    if (v.vst)
    {
        EMU_TRAFFIC_MANUAL("td X, U, uhat (plain pointers)", P::NXV + P::NUV + 3, false);
        const double *Xb = X + size_t(k) * NX, *Ub = U + size_t(k) * NU;
#pragma unroll
        for (int j = 0; j < P::NXV; j++)
            st[L::F_WBAR + j] = Xb[P::XMAP[j]];
#pragma unroll
        for (int j = 0; j < P::NUV; j++)
            st[L::F_WBAR + P::NXV + j] = Ub[P::UMAP[j]];
        for (int j = 0; j < 3; j++)
            st[L::F_UHAT + j] = uhat[size_t(k) * 3 + j];
        // presolved variables (the table's equalTo rows): x_init at the first node, x_final components / zero inputs at the last
        // (zero-order hold: final-input equalities at node K-2, and the non-existent inputs of node K-1 pinned to 0)
#pragma unroll
        for (int j = 0; j < L::NVU; j++)
        {
            if (k == 0 && (P::FIXED_FIRST & (1u << j)))
                st[L::F_W + j] = j < P::NXV ? ip[IP_XINIT + P::XMAP[j < P::NXV ? j : 0]] : 0.;
            if (k == K - 1 && (L::FIX_LAST & (1u << j)))
                st[L::F_W + j] = j < P::NXV ? ip[IP_XFINAL + P::XMAP[j < P::NXV ? j : 0]] : 0.;
            if (k == K - 2 && (L::FIX_PRE & (1u << j)))
                st[L::F_W + j] = 0.;
        }
        Dcount += __builtin_popcount(v.act);
        // identity scalings
        if (!warm)
        {
            for (int i = 0; i < L::NCONES; i++)
                st[L::F_ETA + i] = 1.;
    #pragma unroll
        for (int cix = 0; cix < L::NCONES; cix++)
                st[L::F_WB + L::CONE_OFF.v[cix]] = 1.;
        }
    }
    if (v.vsg)
        Dcount += 2 * L::NL;
    {
        const double dsum = wave_sum(double(Dcount));
        it.D = int(dsum + 0.5) + 3;
    }
    it.bad = 0;
    storePriv(gp, g);
    storePriv(ip_, it);
    WAVE_SYNC();
}

// ---- ECOS init, primal part: rhs of  min ||x||^2 + ||s||^2  s.t. equalities (W = I) ----
template <class P>
PHASE_FN void phInitPrimalRhs(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE("phInitPrimalRhs");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const Views v = makeViews<P>(c);
    const SV &st = v.st, &stN = v.stN, &sg = v.sg, &dy = v.dy;
    const double *ip = c.ip;
    Glob g = loadPriv(gp);
    Iter it = loadPriv(ip_);
    prepareFactor<P>(c, true, g);
    Rhs b;
    if (v.vst)
    {
        double r[L::NS], gw[NV], gdl;
        saff<P>(ip, v.act, st + L::F_W, st[L::F_DL], st + L::F_WBAR, st + L::F_UHAT, r);
        LTmul<P>(ip, v.fm, r, st + L::F_UHAT, gw, &gdl);
        for (int j = 0; j < NV; j++)
            st[L::F_BXW + j] = -gw[j];
        st[L::F_BXD] = -gdl;
    }
    if (v.vsg)
    {
        double res[L::NL];
        dynResF<P>(dy, st + L::F_W, stN + L::F_W, sg + G_NU * L::NL, g.sig, res);
        for (int i = 0; i < L::NL; i++)
        {
            sg[G_BXNU * L::NL + i] = 0.;
            sg[G_BXNUB * L::NL + i] = 0.;
            sg[G_BY * L::NL + i] = -res[i];
        }
    }
    b.s = -((g.sig - 0.001) + (g.sig - it.sigbar));
    b.ds = -(0.5 * (0.5 + 0.5 * g.dsg) - 0.5 * (0.5 - 0.5 * g.dsg));
    b.n1 = 0.;
    b.rhs3 = g.n1;
    it.bts = kktPrep<P>(c, true, g, b);
    it.b = b;
    storePriv(gp, g);
    storePriv(ip_, it);
    WAVE_SYNC();
}
template <class P>
PHASE_FN void phInitPrimalFinish(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE("phInitPrimalFinish");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    Glob g = loadPriv(gp);
    const Iter it = loadPriv(ip_);
    borderSchur<P>(c, g);
    kktFinish<P>(c, true, g, it.b, it.bts);
    applyPrimalStep<P>(c, g, 1.);
    WAVE_SYNC();
    evalAllSaff<P>(c, g, L::F_S, G_S1, G_S2, g.ss, g.s3, g.sc3);
    bring2cone<P>(c, it.gamma, L::F_S, G_S1, G_S2, g.ss, g.s3, g.sc3);
    storePriv(gp, g);
    WAVE_SYNC();
}
// ---- ECOS init, dual part: H x' + A'y = -c ; z = -L x' ----
template <class P>
PHASE_FN void phInitDualRhs(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE("phInitDualRhs");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const Views v = makeViews<P>(c);
    const SV &st = v.st, &sg = v.sg;
    Glob g = loadPriv(gp);
    Iter it = loadPriv(ip_);
    Rhs b;
    if (v.vst)
    {
        for (int j = 0; j < NV; j++)
            st[L::F_BXW + j] = 0.;
        st[L::F_BXD] = -it.wtrx;
    }
    if (v.vsg)
        for (int i = 0; i < L::NL; i++)
        {
            sg[G_BXNU * L::NL + i] = 0.;
            sg[G_BXNUB * L::NL + i] = 0.;
            sg[G_BY * L::NL + i] = 0.;
        }
    b.s = -it.w_t;
    b.ds = -it.w_trt;
    b.n1 = -it.w_vc;
    b.rhs3 = 0.;
    it.bts = kktPrep<P>(c, true, g, b);
    it.b = b;
    storePriv(gp, g);
    storePriv(ip_, it);
    WAVE_SYNC();
}
template <class P>
PHASE_FN void phInitDualFinish(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE("phInitDualFinish");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const Views v = makeViews<P>(c);
    const int K = v.K;
    const SV &st = v.st, &stN = v.stN, &sg = v.sg, &dy = v.dy;
    const double *ip = c.ip;
    Glob g = loadPriv(gp);
    Iter it = loadPriv(ip_);
    kktFinish<P>(c, true, g, it.b, it.bts);
    if (v.vst)
    {
        double t[L::NS];
        Lmul<P>(ip, v.act, st + L::F_DW, st[L::F_DDL], st + L::F_UHAT, t);
        for (int i = 0; i < L::NS; i++)
            st[L::F_Z + i] = -t[i];
    }
    if (v.vsg)
        for (int i = 0; i < L::NL; i++)
        {
            sg[G_LAM * L::NL + i] = sg[G_DLAM * L::NL + i];
            const double dnu = sg[G_DNU * L::NL + i], dnub = sg[G_DNUB * L::NL + i];
            sg[G_Z1 * L::NL + i] = -(dnub - dnu);
            sg[G_Z2 * L::NL + i] = -(dnub + dnu);
        }
    g.zs = -g.dsig;
    g.z3 = g.dz3;
    g.zc3[0] = -0.5 * g.ddsg;
    g.zc3[1] = 0.5 * g.ddsg;
    g.zc3[2] = -g.dsig;
    bring2cone<P>(c, it.gamma, L::F_Z, G_Z1, G_Z2, g.zs, g.z3, g.zc3);
    storePriv(gp, g);
    storePriv(ip_, it);
    WAVE_SYNC();
}
// ---- warm start: slacks re-evaluated on the new data and pushed theta into the interior, duals likewise ----
template <class P>
PHASE_FN void phWarmInit(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE("phWarmInit");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    Glob g = loadPriv(gp);
    const double theta = 1e-2;
    evalAllSaff<P>(c, g, L::F_S, G_S1, G_S2, g.ss, g.s3, g.sc3);
    WAVE_SYNC();
    shiftToCone<P>(c, theta, L::F_S, G_S1, G_S2, g.ss, g.s3, g.sc3);
    shiftToCone<P>(c, theta, L::F_Z, G_Z1, G_Z2, g.zs, g.z3, g.zc3);
    storePriv(gp, g);
    WAVE_SYNC();
}
// ---- data norms for the termination test (ECOS-style scaling) ----
template <class P>
PHASE_FN void phDataNorms(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_, int ddSameIn)
{
    EMU_PHASE("phDataNorms");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const bool dd_same = uniformInt(ddSameIn) != 0; // the norm of the dynamics' constant part is that of the previous solve of this workspace (gsave[13])
    const Views v = makeViews<P>(c);
    const int K = v.K;
    const SV &st = v.st, &stN = v.stN, &dy = v.dy;
    const double *ip = c.ip;
    Iter it = loadPriv(ip_);
    // ---- data norms for the termination test ----
    {
        double resx0 = sqrt(K * it.wtrx * it.wtrx + it.w_t * it.w_t + it.w_trt * it.w_trt + it.w_vc * it.w_vc);
        resx0 = resx0 > 1. ? resx0 : 1.;
        double nb = 0., nh = 0.;
        double w0[NV], w1[NV];
        for (int j = 0; j < NV; j++)
            w0[j] = w1[j] = 0.;
        if (v.vst)
        {
            for (int j = 0; j < NV; j++)
                w0[j] = (v.fm & (1u << j)) ? double(st[L::F_W + j]) : 0.;
            double r[L::NS];
            saff<P>(ip, v.act, w0, 0., st + L::F_WBAR, st + L::F_UHAT, r);
            for (int i = 0; i < L::NS; i++)
                nh += r[i] * r[i];
        }
        if (v.vsg && !dd_same)
        {
            const unsigned fmn = L::fixedMask(v.k + 1, K);
            for (int j = 0; j < NV; j++)
                w1[j] = (fmn & (1u << j)) ? double(stN[L::F_W + j]) : 0.;
            double zero[L::NL], res[L::NL];
            for (int i = 0; i < L::NL; i++)
                zero[i] = 0.;
            dynResF<P>(dy, w0, w1, zero, 0., res);
            for (int i = 0; i < L::NL; i++)
                nb += res[i] * res[i];
        }
        nb = wave_sum(nb);
        nh = wave_sum(nh) + 0.001 * 0.001 + 0.25 + 0.25 + it.sigbar * it.sigbar;
        it.resx0 = resx0;
        // (same data, same pinned variables -> the same number, bit for bit: kept in the warm-start block instead of 128 KB of the copy read again)
        it.resy0 = dd_same ? double(c.gsave[13]) : (sqrt(nb) > 1. ? sqrt(nb) : 1.);
        if (!dd_same && c.lane == 0)
            c.gsave[13] = it.resy0;
        it.resz0 = sqrt(nh) > 1. ? sqrt(nh) : 1.;
    }
    storePriv(ip_, it);
    WAVE_SYNC();
}


// batched field access: all loads (stores) of a group are issued back to back; buffer stores may alias buffer
// loads as far as the compiler knows, so every phase is written as  load group -> compute -> store group
template <int N>
__device__ inline void ldf(const SV &r, int f, double (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++)
        v[i] = r[f + i];
}
template <int N>
__device__ inline void stf(const SV &r, int f, const double (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++)
        r[f + i] = v[i];
}

// =====================================================================================================
// Segment rows (virtual control nu_k, its bound nu_b, the LP pair nu_b -/+ nu >= 0 and the multiplier lam of the dynamics row).
// Round 4: everything a lane phase needs of a row is RECOMPUTED from the seven state fields (nu, nu_b, s1, z1, s2, z2, lam), the
// product fields of the predictor (ds_aff dz_aff of the two LP cones) and wave-uniform scalars, with ONE set of expressions
// (segRhs / segDir below) shared by the right-hand-side, direction and update phases.  Until round 3 the residuals (rz1, rz2, rxnu,
// rxnub), the right-hand-side pieces (t1, t2, bnb, btn) and the seven final directions were stored by one phase and re-read by the
// next: 103 field accesses per row and interior-point iteration, now 71 -- the kernel is bound by the bytes it moves,
// and the workspace streams from HBM, so a field that is not stored is a field that is not written AND not read back.
// =====================================================================================================
template <int N>
struct SegState
{
    double nu[N], nub[N], s1[N], z1[N], s2[N], z2[N], lam[N];
};
// this lane's view of the three most-travelled segment fields (ipm_kernel.h: SegLdsSlot): rows [I0, I0 + N) of slot SLOT -- in LDS, or
// (SegFieldsInWorkspace<P>: kernels with more than two wavefronts per SIMD) their home in the field-major segment record of the workspace
struct SegLds
{
    LDSP double *p; // + segment index
    int pitch;
    SV sg;          // the segment record of the same segment (workspace-resident variant)
};
template <int SLOT>
constexpr int segLdsField()
{
    return SLOT == SL_LAM ? int(G_LAM) : SLOT == SL_NU ? int(G_NU) : int(G_NUB);
}
template <class P>
__device__ inline SegLds makeSegLds(const Ctx &c, int seg)
{
    return SegLds{c.segl + seg, c.pitch, makeSV(c.sg, (G_NFIELDS * Lay<P>::NL), unsigned(seg), c.pitch)};
}
// one element (direct access of phResiduals)
template <class P, int SLOT>
__device__ inline double segLdsGet(const SegLds &l, int i)
{
    if constexpr (SegInLds<P>::value)
        return l.p[(SLOT * Lay<P>::NL + i) * l.pitch];
    else
        return l.sg[segLdsField<SLOT>() * Lay<P>::NL + i];
}
template <class P, int SLOT, int I0, int N>
__device__ inline void ldl(const SegLds &l, double (&v)[N])
{
    if constexpr (SegInLds<P>::value)
    {
#pragma unroll
        for (int i = 0; i < N; i++)
            v[i] = l.p[(SLOT * Lay<P>::NL + I0 + i) * l.pitch];
    }
    else
        ldf<N>(l.sg, segLdsField<SLOT>() * Lay<P>::NL + I0, v);
}
template <class P, int SLOT, int I0, int N>
__device__ inline void stl(const SegLds &l, const double (&v)[N])
{
    if constexpr (SegInLds<P>::value)
    {
#pragma unroll
        for (int i = 0; i < N; i++)
            l.p[(SLOT * Lay<P>::NL + I0 + i) * l.pitch] = v[i];
    }
    else
        stf<N>(l.sg, segLdsField<SLOT>() * Lay<P>::NL + I0, v);
}
template <class P, int I0, int N>
__device__ inline void ldSegState(const SV &sg, const SegLds &sl, SegState<N> &q)
{
    using L = Lay<P>;
    ldf<N>(sg, G_S1 * L::NL + I0, q.s1);
    ldf<N>(sg, G_Z1 * L::NL + I0, q.z1);
    ldf<N>(sg, G_S2 * L::NL + I0, q.s2);
    ldf<N>(sg, G_Z2 * L::NL + I0, q.z2);
    ldl<P, SL_NU, I0, N>(sl, q.nu);
    ldl<P, SL_NUB, I0, N>(sl, q.nub);
    ldl<P, SL_LAM, I0, N>(sl, q.lam);
}
// right-hand-side pieces of the eliminated LP pair of one row (pass 0: affine, pass 1: corrector with the predictor's products p1, p2)
struct SegRhsRow
{
    double rz1, rz2, t1, t2, bnb, btn, einv, qv, d1, d2, dinv;
    double is1, iz1, is2, iz2; // reciprocals of the slacks and duals of the LP pair
};
// SEG_FAST_RCP: every quotient of a row goes through ONE reciprocal per slack / dual (v_rcp_f64 + two Newton steps, common.h: fastRcp:
// 5 instructions, within 1 ulp) and multiplications, instead of an IEEE division (11 instructions, two of them quarter rate) each: a
// row takes 8 .. 12 divisions per phase, 44 per interior-point iteration, i.e. ~600 per lane.  The operands are slacks and duals of an
// interior point: positive, normal.
#ifndef RES_PREVLANE_MEMORY
#define RES_PREVLANE_MEMORY 0
#endif
#ifndef SEG_FAST_RCP
#define SEG_FAST_RCP 1
#endif
__device__ inline double segRcp(double x)
{
#if SEG_FAST_RCP
    return fastRcp(x);
#else
    return 1. / x;
#endif
}
template <int PASS>
__device__ inline SegRhsRow segRhsRow(double nu, double nub, double s1, double z1, double s2, double z2, double lam, double p1, double p2, double om,
                                      double sigmu, double z3, double dz3)
{
    SegRhsRow o;
    // residuals of the row at the current iterate (what phResiduals evaluates for the termination test)
    o.rz1 = s1 - (nub - nu);
    o.rz2 = s2 - (nub + nu);
    const double rxnu = -lam + z1 - z2, rxnub = -z1 - z2 + z3;
#if SEG_FAST_RCP
    o.is1 = segRcp(s1);
    o.iz1 = segRcp(z1);
    o.is2 = segRcp(s2);
    o.iz2 = segRcp(z2);
    double c1 = 0., c2 = 0.;
    if (PASS)
    {
        c1 = (sigmu - p1) * o.is1;
        c2 = (sigmu - p2) * o.is2;
    }
    o.d1 = z1 * o.is1;
    o.d2 = z2 * o.is2;
#else
    double c1 = 0., c2 = 0.;
    if (PASS)
    {
        c1 = (sigmu - p1) / s1;
        c2 = (sigmu - p2) / s2;
    }
    o.d1 = z1 / s1;
    o.d2 = z2 / s2;
#endif
    o.t1 = o.d1 * om * o.rz1 - z1 + c1;
    o.t2 = o.d2 * om * o.rz2 - z2 + c2;
    const double bxnu = -om * rxnu + (-o.t1 + o.t2);
    const double bxnub = -om * rxnub + (o.t1 + o.t2);
    // E^-1, q and 1 / (d1 + d2) of the eliminated LP pair
#if SEG_FAST_RCP
    const double r1 = s1 * o.iz1, r2 = s2 * o.iz2;
    o.einv = 0.25 * (r1 + r2);
    o.qv = (r1 - r2) * segRcp(r1 + r2);
    o.dinv = segRcp(o.d1 + o.d2);
#else
    const double r1 = s1 / z1, r2 = s2 / z2;
    o.einv = 0.25 * (r1 + r2);
    o.qv = (r1 - r2) / (r1 + r2);
    o.dinv = 1. / (o.d1 + o.d2);
#endif
    o.bnb = bxnub - dz3;
    o.btn = bxnu - o.qv * o.bnb;
    return o;
}
// Newton direction of one row from the block solve's multiplier direction vl (- border column * dsigma)
struct SegDirRow
{
    double dlam, dnu, dnub, dz1, ds1, dz2, ds2;
};
__device__ inline SegDirRow segDirRow(const SegRhsRow &r, double vl, double bcl, double om, double dsig)
{
    SegDirRow o;
    o.dlam = vl - bcl * dsig;
    o.dnu = r.einv * (o.dlam + r.btn);
    o.dnub = r.bnb * r.dinv - r.qv * o.dnu;
    const double L1v = o.dnub - o.dnu, L2v = o.dnub + o.dnu;
    o.dz1 = -r.d1 * L1v + r.t1;
    o.ds1 = -om * r.rz1 + L1v;
    o.dz2 = -r.d2 * L2v + r.t2;
    o.ds2 = -om * r.rz2 + L2v;
    return o;
}

// ---- residuals, duality gap, termination quantities ----
#ifdef SCPP_HIP_EMU
// (one counter for both instantiations of phResiduals: a function of its own, not statics inside the template)
inline void emuInjectResiduals(int lane, double &pres, double &dres, double &gap)
{
    static int calls[LANES];
    static int inj_n = -2, inj_m = -1; // "n:pres:dres:gap" or "n,m:pres:dres:gap" (two evaluations: the first attempt's and the repeated one's)
    static double inj_v[3];
    if (inj_n == -2)
    {
        const char *e = getenv("SCPP_EMU_INJECT_RES");
        inj_n = -1;
        if (e && sscanf(e, "%d,%d:%lf:%lf:%lf", &inj_n, &inj_m, &inj_v[0], &inj_v[1], &inj_v[2]) != 5)
        {
            inj_m = -1;
            if (sscanf(e, "%d:%lf:%lf:%lf", &inj_n, &inj_v[0], &inj_v[1], &inj_v[2]) != 4)
                inj_n = -1;
        }
    }
    const int call = inj_n >= 0 ? calls[lane & (LANES - 1)]++ : -1;
    if (inj_n >= 0 && (call == inj_n || call == inj_m))
    {
        pres = inj_v[0];
        dres = inj_v[1];
        gap = inj_v[2];
    }
}
#endif
struct ResAcc
{
    double gap, rx, ry, rz, xx, yy, zz, ss, rxs, sumnb;
};
// what the pending step of the previous iteration needs (UPD: phResiduals applies it on the way in, see there)
struct PendingStep
{
    double om, sigmu, z3_old, dz3, dsig, alpha, alpha_d; // alpha: primal step length, alpha_d: dual (IPM_SPLIT_STEPS)
};
template <class P, int I0, int N, bool UPD>
__device__ inline void resSegChunk(const SV &sg, const SegLds &sl, const SV &dyz, const SV &xs, const SV &xsz, double z3, ResAcc &p, const PendingStep &u)
{
    using L = Lay<P>;
    double nu[N], nub[N], s1[N], z1[N], s2[N], z2[N], lam[N], S[N];
    ldf<N>(sg, G_S1 * L::NL + I0, s1);
    ldf<N>(sg, G_Z1 * L::NL + I0, z1);
    ldf<N>(sg, G_S2 * L::NL + I0, s2);
    ldf<N>(sg, G_Z2 * L::NL + I0, z2);
    ldl<P, SL_NU, I0, N>(sl, nu);
    ldl<P, SL_NUB, I0, N>(sl, nub);
    ldl<P, SL_LAM, I0, N>(sl, lam);
    ldf<N>(dyz, L::DY_S + I0, S); // dS/dsigma column: zero for a fixed final time (SCvx), read through the padded view
    if constexpr (UPD)
    {
        // x += alpha dx ; s += alpha ds ; z += alpha dz of the rows' PREVIOUS iteration, applied here instead of in a phase of its own (round 6): the
        // seven state fields are read ONCE for the step and for the residuals of the point it leads to.  The arithmetic is updSegChunk's, statement for
        // statement (the corrector direction recomputed from the state, the predictor's products and the block solve's multiplier direction).
        double vl[N], bcl[N], p1[N], p2[N];
        ldf<N>(xs, L::X_VL + I0, vl);
        ldf<N>(xsz, L::X_BCL + I0, bcl);
        ldf<N>(sg, G_DS1 * L::NL + I0, p1);
        ldf<N>(sg, G_DS2 * L::NL + I0, p2);
        LOADS_ISSUED();
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            const SegRhsRow r = segRhsRow<1>(nu[i], nub[i], s1[i], z1[i], s2[i], z2[i], lam[i], p1[i], p2[i], u.om, u.sigmu, u.z3_old, u.dz3);
            const SegDirRow d = segDirRow(r, vl[i], bcl[i], u.om, u.dsig);
            nu[i] = nu[i] + u.alpha * d.dnu;
            nub[i] = nub[i] + u.alpha * d.dnub;
            lam[i] = lam[i] + u.alpha_d * d.dlam;
            s1[i] = s1[i] + u.alpha * d.ds1;
            z1[i] = z1[i] + u.alpha_d * d.dz1;
            s2[i] = s2[i] + u.alpha * d.ds2;
            z2[i] = z2[i] + u.alpha_d * d.dz2;
        }
        stl<P, SL_NU, I0, N>(sl, nu);
        stl<P, SL_NUB, I0, N>(sl, nub);
        stl<P, SL_LAM, I0, N>(sl, lam);
        stf<N>(sg, G_S1 * L::NL + I0, s1);
        stf<N>(sg, G_Z1 * L::NL + I0, z1);
        stf<N>(sg, G_S2 * L::NL + I0, s2);
        stf<N>(sg, G_Z2 * L::NL + I0, z2);
    }
    else
        LOADS_ISSUED();
    // (the row residuals are not stored: the right-hand-side / direction phases recompute them from the state, segRhsRow)
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        const double r1 = s1[i] - (nub[i] - nu[i]);
        const double r2 = s2[i] - (nub[i] + nu[i]);
        const double rnu = -lam[i] + z1[i] - z2[i];
        const double rnub = -z1[i] - z2[i] + z3;
        p.sumnb += nub[i];
        p.gap += s1[i] * z1[i] + s2[i] * z2[i];
        p.rz += r1 * r1 + r2 * r2;
        p.zz += z1[i] * z1[i] + z2[i] * z2[i];
        p.ss += s1[i] * s1[i] + s2[i] * s2[i];
        p.yy += lam[i] * lam[i];
        p.rx += rnu * rnu + rnub * rnub;
        p.xx += nu[i] * nu[i] + nub[i] * nub[i];
        p.rxs += S[i] * lam[i];
    }
}
// UPD (round 6): the step of the previous iteration (phUpdate's work) is applied on the way into the residual pass -- the seven segment fields and W, delta,
// s, z of a stage are read once for the step AND for the residuals of the new point instead of being stored by one phase and loaded by the next
// (-47 KB of 1.92 MB per iteration, tools/traffic_table.py).  Same operations in the same order on every entry: bitwise the two phases.
template <class P, bool UPD>
PHASE_FN void phResiduals(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE(UPD ? "phResiduals<update>" : "phResiduals");
    using L = Lay<P>;
    constexpr int NX = P::NX, NU = P::NU;
    const Ctx c = uniformCtx(cin);
    const Views v = makeViews<P>(c);
    const int k = v.k;
    const SV &st = v.st, &stN = v.stN, &sg = v.sg, &sgP = v.sgP, &dy = v.dy, &dyP = v.dyP;
    const double *ip = c.ip;
    const unsigned fm = v.fm;
    PendingStep u{0., 0., 0., 0., 0., 0., 0.};
    if constexpr (UPD)
    {
        // the wave-uniform rows of the step (phUpdate's tail), first: everything below reads the new point
        Glob g = loadPriv(gp);
        const double sigma_c = ip_->sigma_c;
        u.alpha = ip_->alpha;
        u.alpha_d = ip_->alpha_d;
        u.om = 1. - sigma_c;
        u.sigmu = sigma_c * double(ip_->mu);
        u.z3_old = g.z3;
        u.dz3 = g.dz3;
        u.dsig = g.dsig;
        const double alpha = u.alpha, alpha_d = u.alpha_d;
        g.sig += alpha * g.dsig;
        g.dsg += alpha * g.ddsg;
        g.n1 += alpha * g.dn1;
        g.ss += alpha * g.dss;
        g.zs += alpha_d * g.dzs;
        g.s3 += alpha * g.ds3;
        g.z3 += alpha_d * g.dz3;
        for (int i = 0; i < 3; i++)
        {
            g.sc3[i] += alpha * g.dsc3[i];
            g.zc3[i] += alpha_d * g.dzc3[i];
        }
        PUT_BEGIN();
        PUT(gp, g, sig);
        PUT(gp, g, dsg);
        PUT(gp, g, n1);
        PUT(gp, g, ss);
        PUT(gp, g, zs);
        PUT(gp, g, s3);
        PUT(gp, g, z3);
        PUT(gp, g, sc3);
        PUT(gp, g, zc3);
        PUT_END();
    }
    const double g_z3 = gp->z3, g_sig = gp->sig, it_wtrx = ip_->wtrx;
    const bool scvx = scvxMode(ip);
    const SV stz = padView(v.st, scvx);
    const SV dyz = padView(v.dy, scvx); // for the S column only
    // LDS-resident segment fields of this lane's segment and of the previous one (lanes without a segment read segment 0, masked)
    const SegLds sl = makeSegLds<P>(c, v.vsg ? k : 0), slP = makeSegLds<P>(c, (k > 0 && k < v.K) ? k - 1 : 0);
    ResAcc p;
    p.gap = p.rx = p.ry = p.rz = p.xx = p.yy = p.zz = p.ss = p.rxs = p.sumnb = 0.;
    double p_dl = 0.;
    if (v.vsg)
    {
        const SV xsz = padView(v.xs, scvx);
        forSegChunks<P, chunkFor<P>(IPM_RES_CHUNK, IPM_RES_CHUNK_W)>(
            [&](auto i0, auto n) { resSegChunk<P, decltype(i0)::value, decltype(n)::value, UPD>(sg, sl, dyz, v.xs, xsz, g_z3, p, u); });
    }
    double x0[NV], gw[NV], gdl = 0., dl = 0.;
    if (v.vst)
    {
        {
            // rz = s - saff<P>(x) ; gap ; L'z
            double wbar[NV], uh[3], sa[L::NS], sv[L::NS], zv[L::NS];
            ldf<NV>(st, L::F_W, x0);
            ldf<NV>(st, L::F_WBAR, wbar);
            ldf<3>(st, L::F_UHAT, uh);
            dl = st[L::F_DL];
            if constexpr (UPD)
            {
                // the stage's share of the step (phUpdate's arithmetic: w += alpha dw on the free variables, delta, s += alpha ds), kept in registers.
                // The directions arrive in pieces (W ; two halves of ds ; two halves of dz): all of them next to s, z, rz and W at once is 40 doubles
                // more than a lane has (first form of this pass: 444 bytes of scratch per lane against 128, and 1.1 % SLOWER than the two phases).
                constexpr int H0 = (L::NS + 1) / 2, H1 = L::NS - H0;
                {
                    double dw[NV];
                    ldf<NV>(st, L::F_DW, dw);
                    const double ddl = st[L::F_DDL];
                    LOADS_ISSUED();
#pragma unroll
                    for (int j = 0; j < NV; j++)
                        x0[j] = x0[j] + ((fm & (1u << j)) ? 0. : u.alpha * dw[j]);
                    dl = dl + u.alpha * ddl;
                }
                ldPad<L::NS, P::NXV>(st, stz, L::F_S, sv);
                {
                    double d[H0];
                    ldPadPart<P::NXV, 0, H0>(st, stz, L::F_DS, d);
                    LOADS_ISSUED();
#pragma unroll
                    for (int i = 0; i < H0; i++)
                        sv[i] = sv[i] + u.alpha * d[i];
                }
                {
                    double d[H1];
                    ldPadPart<P::NXV, H0, H1>(st, stz, L::F_DS, d);
                    LOADS_ISSUED();
#pragma unroll
                    for (int i = 0; i < H1; i++)
                        sv[H0 + i] = sv[H0 + i] + u.alpha * d[i];
                }
                ldPad<L::NS, P::NXV>(st, stz, L::F_Z, zv); // requested ahead of the stores below (a load behind a store waits for it)
                stf<NV>(st, L::F_W, x0);
                st[L::F_DL] = dl;
                stPad<L::NS, P::NXV>(st, stz, L::F_S, sv);
            }
            else
            {
                ldPad<L::NS, P::NXV>(st, stz, L::F_S, sv);
                LOADS_ISSUED();
            }
            saff<P>(ip, v.act, x0, dl, wbar, uh, sa);
#pragma unroll
            for (int i = 0; i < L::NS; i++)
            {
                sa[i] = sv[i] - sa[i];
                p.rz += sa[i] * sa[i];
                p.ss += sv[i] * sv[i];
            }
            if constexpr (UPD)
            {
                constexpr int H0 = (L::NS + 1) / 2, H1 = L::NS - H0;
                stPad<L::NS, P::NXV>(st, stz, L::F_RZ, sa);
                {
                    double d[H0];
                    ldPadPart<P::NXV, 0, H0>(st, stz, L::F_DZ, d);
                    LOADS_ISSUED();
#pragma unroll
                    for (int i = 0; i < H0; i++)
                        zv[i] = zv[i] + u.alpha_d * d[i];
                }
                {
                    double d[H1];
                    ldPadPart<P::NXV, H0, H1>(st, stz, L::F_DZ, d);
                    LOADS_ISSUED();
#pragma unroll
                    for (int i = 0; i < H1; i++)
                        zv[H0 + i] = zv[H0 + i] + u.alpha_d * d[i];
                }
                stPad<L::NS, P::NXV>(st, stz, L::F_Z, zv);
            }
            else
            {
                ldPad<L::NS, P::NXV>(st, stz, L::F_Z, zv); // requested ahead of the stores below (a load behind a store waits for it)
                LOADS_ISSUED();
                stPad<L::NS, P::NXV>(st, stz, L::F_RZ, sa);
            }
#pragma unroll
            for (int i = 0; i < L::NS; i++)
            {
                p.gap += sv[i] * zv[i];
                p.zz += zv[i] * zv[i];
            }
            LTmul<P>(ip, fm, zv, uh, gw, &gdl);
        }
    }
    if constexpr (UPD)
    {
        // the neighbours' new W (stN below) and lambda (LDS) were written by other lanes of this wavefront a moment ago: a function boundary stood
        // between the two phases (s_waitcnt vmcnt(0) is part of the call ABI); here the wait is explicit (outside the lane-dependent branches: the
        // emulator's lanes are fibers that meet at synchronisation points)
#ifndef SCPP_HIP_EMU
        __builtin_amdgcn_s_waitcnt(0);
#endif
        WAVE_SYNC();
    }
    if (v.vst)
    {
        const double rxd = (ip[IP_SCVX] != 0.) ? 0. : it_wtrx - gdl; // SCvx: delta_k is not a variable
        // r = -L'z + M_k' lam_k + N_{k-1}' lam_{k-1}   and   res = x_{k+1} - A x_k - B u_k - C u_{k+1} - S sigma - nu - Z
        // in ONE pass over the field-major copy of (A,B,C): the loads of a row are issued together
        constexpr int NXV = P::NXV, NUV = P::NUV;
        double acc[NV], u1[NUV], resv[L::NL];
#pragma unroll
        for (int j = 0; j < NV; j++)
            acc[j] = 0.;
        ldf<NUV>(stN, L::F_W + NXV, u1);
        const double mk = v.vsg ? 1. : 0., mp = k > 0 ? 1. : 0.;
#pragma unroll
        for (int i = 0; i < L::NL; i++)
        {
            double ra[NXV], rb[NUV], rc[NUV], rcp[NUV];
            sfor<NXV>([&](auto jt) { ra[decltype(jt)::value] = dy[L::DY_A + i * NX + P::XMAP[decltype(jt)::value]]; });
            sfor<NUV>([&](auto jt) { rb[decltype(jt)::value] = dy[L::DY_B + i * NU + P::UMAP[decltype(jt)::value]]; });
            sfor<NUV>([&](auto jt) { rc[decltype(jt)::value] = dy[L::DY_C + i * NU + P::UMAP[decltype(jt)::value]]; });
            // C of the PREVIOUS segment = what lane k - 1 just loaded as its own C row: one wavefront shift instead of a second load
            // (the fiber emulator cannot exchange inside a divergent region: it reads the same value from memory)
            // (-DRES_PREVLANE_MEMORY=1 builds the memory-load form for the device too: tests/tools/lib_equal.py compares the two libraries bitwise on
            // the GPU -- the shift is a data path the CPU suite never runs)
#if defined(SCPP_HIP_EMU) || RES_PREVLANE_MEMORY
            sfor<NUV>([&](auto jt) { rcp[decltype(jt)::value] = dyP[L::DY_C + i * NU + P::UMAP[decltype(jt)::value]]; });
#else
            (void)dyP;
            sfor<NUV>([&](auto jt) { rcp[decltype(jt)::value] = prevLane(rc[decltype(jt)::value]); });
#endif
            const double l = mk * segLdsGet<P, SL_LAM>(sl, i), lp = mp * segLdsGet<P, SL_LAM>(slP, i);
            const int xi = L::XINV.v[i]; // stage variable of state i, -1: pinned
            double rr = (xi >= 0 ? double(stN[L::F_W + (xi >= 0 ? xi : 0)]) : 0.) - dyz[L::DY_S + i] * g_sig - segLdsGet<P, SL_NU>(sl, i) - dy[L::DY_Z + i];
#pragma unroll
            for (int j = 0; j < NXV; j++)
            {
                acc[j] -= ra[j] * l;
                rr -= ra[j] * x0[j];
            }
#pragma unroll
            for (int j = 0; j < NUV; j++)
            {
                acc[NXV + j] -= rb[j] * l + rcp[j] * lp;
                rr -= rb[j] * x0[NXV + j] + rc[j] * u1[j];
            }
            if (xi >= 0)
                acc[xi >= 0 ? xi : 0] += lp;
            resv[i] = rr;
            p.ry += v.vsg ? rr * rr : 0.;
        }
        p.rx += rxd * rxd;
        p.xx += dl * dl;
        p_dl += dl;
        double rxw[NV];
#pragma unroll
        for (int j = 0; j < NV; j++)
        {
            const bool fx = fm & (1u << j);
            const double r = -gw[j] + (fx ? 0. : acc[j]);
            rxw[j] = r;
            p.rx += fx ? 0. : r * r;
            p.xx += fx ? 0. : x0[j] * x0[j];
        }
        st[L::F_RXD] = rxd;
        stf<NV>(st, L::F_RXW, rxw);
        if (v.vsg)
            stf<L::NL>(sg, G_RY * L::NL, resv);
    }
    p.gap = waveSumDpp(p.gap);
    p.rx = waveSumDpp(p.rx);
    p.ry = waveSumDpp(p.ry);
    p.rz = waveSumDpp(p.rz);
    p.xx = waveSumDpp(p.xx);
    p.yy = waveSumDpp(p.yy);
    p.zz = waveSumDpp(p.zz);
    p.ss = waveSumDpp(p.ss);
    p.rxs = waveSumDpp(p.rxs);
    p.sumnb = waveSumDpp(p.sumnb);
    p_dl = waveSumDpp(p_dl);
    const Glob g = reloadPriv(gp);
    Iter it = reloadPriv(ip_);
    const double pres_before = it.pres; // the previous iteration's (meaningful while a backup of THIS solve exists: bk_valid is cleared at the solve's start)
    const double sas = g.sig - 0.001, sa3 = g.n1 - p.sumnb;
    const double sac[3] = {0.5 + 0.5 * g.dsg, 0.5 - 0.5 * g.dsg, g.sig - g.sigbar};
    it.rzs = g.ss - sas;
    it.rz3 = g.s3 - sa3;
    for (int i = 0; i < 3; i++)
        it.rzc[i] = g.sc3[i] - sac[i];
    it.rxs = it.w_t - g.zs - g.zc3[2] - p.rxs;
    it.rxds = it.w_trt - 0.5 * g.zc3[0] + 0.5 * g.zc3[1];
    it.rxn1 = it.w_vc - g.z3;
    double gap = p.gap + g.ss * g.zs + g.s3 * g.z3;
    double nrz = p.rz + it.rzs * it.rzs + it.rz3 * it.rz3;
    double nzz = p.zz + g.zs * g.zs + g.z3 * g.z3;
    double nss = p.ss + g.ss * g.ss + g.s3 * g.s3;
    for (int i = 0; i < 3; i++)
    {
        gap += g.sc3[i] * g.zc3[i];
        nrz += it.rzc[i] * it.rzc[i];
        nzz += g.zc3[i] * g.zc3[i];
        nss += g.sc3[i] * g.sc3[i];
    }
    const double nrx = p.rx + it.rxs * it.rxs + it.rxds * it.rxds + it.rxn1 * it.rxn1;
    const double nxx = p.xx + g.sig * g.sig + g.dsg * g.dsg + g.n1 * g.n1;
    it.gap = gap;
    it.mu = gap / it.D;
    it.pcost = it.w_t * g.sig + it.w_trt * g.dsg + it.w_vc * g.n1 + it.wtrx * p_dl;
    {
        const double nx_ = sqrt(nxx), ny_ = sqrt(p.yy), nz_ = sqrt(nzz), ns_ = sqrt(nss);
        const double d1 = it.resy0 + nx_ > 1. ? it.resy0 + nx_ : 1.;
        const double d2 = it.resz0 + nx_ + ns_ > 1. ? it.resz0 + nx_ + ns_ : 1.;
        const double pa = sqrt(p.ry) / d1, pb = sqrt(nrz) / d2;
        it.pres = pa > pb ? pa : pb;
        const double d3 = it.resx0 + ny_ + nz_ > 1. ? it.resx0 + ny_ + nz_ : 1.;
        it.dres = sqrt(nrx) / d3;
    }
#ifdef SCPP_HIP_EMU
    // Test support (emulator build only): SCPP_EMU_INJECT_RES="n:pres:dres:gap" replaces the termination quantities of the n-th residual evaluation of
    // every instance of the process (counted per lane: the lanes of the emulated wavefront are fibers of one host thread) -- a numerically broken
    // iterate on demand, for the regression tests of the breakdown rules (IPM_BLOWN, IPM_NEG_GAP).
    emuInjectResiduals(c.lane, it.pres, it.dres, it.gap);
#endif
    {
        // keep the iterate if it already meets ECOS's reduced tolerances (returned if the path breaks down later)
        const double apc = fabs(it.pcost) > 1e-300 ? fabs(it.pcost) : 1e-300;
        // ... unless it is BROKEN by the very test the solver's loop applies next (ipmSolveInstance: non-finite, blown up, or the residuals exploded / the
        // gap went negative after a backup existed).  The scalar twin runs that test BEFORE it saves (oracle/structured_ipm.hpp); here the save comes
        // first, and until round 5 a blown-up iterate -- pres = 0 in these relative measures, gap = -2.3e303 < 5e-5 -- OVERWROTE the good backup it
        // was about to be replaced by: instance 8392 returned inputs of 1e154 N with status 0 (bench line: solver_failures 1 of 32768).
        const bool broken = !(it.pres == it.pres) || !(it.dres == it.dres) || !(it.gap == it.gap) || fabs(it.pres) > 1e300 || fabs(it.dres) > 1e300 ||
                            ipmGapBroken(it.gap) || fabs(it.pcost) > IPM_BLOWN ||
                            (it.bk_valid != 0 && (it.pres > 500. * pres_before || it.gap < 0.));
        const bool inacc = !broken && it.pres < 1e-4 && it.dres < 1e-4 && (it.gap < 5e-5 || it.gap / apc < 5e-5);
        if (inacc)
        {
            if (v.vst)
            {
                double xw[NV + 1];
                ldf<NV + 1>(st, L::F_W, xw); // W[16], delta
                stf<NV + 1>(st, L::F_WBK, xw);
            }
            it.bk_sig = g.sig;
            it.bk_dsg = g.dsg;
            it.bk_n1 = g.n1;
            it.bk_valid = 1;
        }
        PUT_BEGIN();
        PUT(ip_, it, rzs);
        PUT(ip_, it, rz3);
        PUT(ip_, it, rzc);
        PUT(ip_, it, rxs);
        PUT(ip_, it, rxds);
        PUT(ip_, it, rxn1);
        PUT(ip_, it, gap);
        PUT(ip_, it, mu);
        PUT(ip_, it, pcost);
        PUT(ip_, it, pres);
        PUT(ip_, it, dres);
        if (inacc)
        {
            PUT(ip_, it, bk_sig);
            PUT(ip_, it, bk_dsg);
            PUT(ip_, it, bk_n1);
            PUT(ip_, it, bk_valid);
        }
        PUT_END();
    }
}

// ---- Nesterov-Todd scalings and the per-stage factorisation inputs ----
template <class P>
PHASE_FN void phScalings(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE("phScalings");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const Views v = makeViews<P>(c);
    const SV &st = v.st;
    const unsigned act = v.act;
    const bool scvx = scvxMode(c.ip);
    const SV stz = padView(v.st, scvx);
    Glob g = loadPriv(gp);
    int bad = 0;
    if (v.vst)
    {
        forEachCone<P>([&](auto ci) {
            constexpr int C = decltype(ci)::value;
            if (act & (1u << C))
                bad |= coneScaling<P, L::coneOff(C), L::coneDim(C)>(st, C, stz);
        });
    }
#ifdef SCPP_HIP_EMU
    if (bad && getenv("SCPP_EMU_DEBUG"))
        printf("[emu] lane %d stage cone scaling failed\n", v.k);
#endif
    if (!cone::nt_scaling(g.sc3, g.zc3, 3, g.seta, g.sw))
    {
        bad = 1;
#ifdef SCPP_HIP_EMU
        if (c.lane == 0 && getenv("SCPP_EMU_DEBUG"))
            printf("[emu] sigma cone scaling failed: s=(%g %g %g) z=(%g %g %g)\n", g.sc3[0], g.sc3[1], g.sc3[2], g.zc3[0], g.zc3[1], g.zc3[2]);
#endif
    }
    bad = waveOrBallot(bad);
    if (!bad)
    {
        cone::applyW(g.seta, g.sw, 3, g.zc3, g.lamC);
        WAVE_SYNC();
        prepareFactor<P>(c, false, g);
    }
    PUT_BEGIN();
    ip_->bad = bad;
    PUT(gp, g, seta);
    PUT(gp, g, sw);
    PUT(gp, g, lamC);
    PUT(gp, g, Hsd);
    PUT(gp, g, Hdd);
    PUT(gp, g, hsig);
    PUT_END();
}

// ---- right-hand side of one Newton system: t = W^-2 rz' + W^-1(lambda \ ds) ; bx = -rx' + L't ; condensation ----
// segment rows [I0, I0+N): LP blocks of nu / nu_b, fused with the condensation (kktPrep) of those rows
template <class P, int I0, int N, int PASS>
__device__ inline void rhsSegChunk(const SV &sg, const SegLds &sl, const SV &xs, double om, double sigmu, double z3, double dz3)
{
    using L = Lay<P>;
    SegState<N> q;
    double ry[N];
    double p1[N], p2[N]; // ds_aff dz_aff of the two LP cones (written by the predictor's direction phase)
    ldSegState<P, I0, N>(sg, sl, q);
    ldf<N>(sg, G_RY * L::NL + I0, ry);
    if (PASS)
    {
        ldf<N>(sg, G_DS1 * L::NL + I0, p1);
        ldf<N>(sg, G_DS2 * L::NL + I0, p2);
    }
    LOADS_ISSUED();
    double rho[N], einv[N];
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        const SegRhsRow r = segRhsRow<PASS>(q.nu[i], q.nub[i], q.s1[i], q.z1[i], q.s2[i], q.z2[i], q.lam[i], PASS ? p1[i] : 0., PASS ? p2[i] : 0., om,
                                            sigmu, z3, dz3);
        const double by = -om * ry[i];
        rho[i] = by + r.einv * r.btn;
        einv[i] = r.einv;
    }
    stf<N>(xs, L::X_RHO + I0, rho);
    if (!PASS)
        stf<N>(xs, L::X_EINV + I0, einv); // E^-1 of the factorisation (one per iteration: the predictor pass writes it)
}
template <class P, int PASS>
PHASE_FN void phRhs(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE(PASS ? "phRhs<1>" : "phRhs<0>");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    constexpr int pass = PASS; // 0: predictor (affine), 1: corrector -- two instantiations, no run-time branches around loads
    const Views v = makeViews<P>(c);
    const SV &st = v.st, &sg = v.sg;
    const double *ip = c.ip;
    const unsigned fm = v.fm, act = v.act;
    const bool scvx = scvxMode(ip);
    const SV stz = padView(v.st, scvx);
    Glob g = loadPriv(gp);
    Iter it = loadPriv(ip_);
    const double sigma_c = pass ? it.sigma_c : 0., mu = it.mu;
    const double om = 1. - sigma_c, sigmu = sigma_c * mu;
    // ---- wave-uniform rows (sigma, delta_sigma, n1 and their cones) ----
    it.tzs = (g.zs / g.ss) * om * it.rzs - g.zs + (pass ? (sigmu - g.dss * g.dzs) / g.ss : 0.);
    {
        double aa[3], b2[3];
        for (int i = 0; i < 3; i++)
            aa[i] = om * it.rzc[i];
        cone::applyWinv2(g.seta, g.sw, 3, aa, b2);
        if (pass == 0)
            for (int i = 0; i < 3; i++)
                it.tzc[i] = b2[i] - g.zc3[i];
        else
        {
            double dsv[3];
            cone::conicProduct(3, g.dsC, g.dzC, dsv);
            for (int i = 0; i < 3; i++)
                dsv[i] = -dsv[i];
            dsv[0] += sigmu;
            cone::conicDivision(3, g.lamC, dsv, dsv);
            for (int i = 0; i < 3; i++)
                dsv[i] -= g.lamC[i];
            cone::applyWinv(g.seta, g.sw, 3, dsv, aa);
            for (int i = 0; i < 3; i++)
                it.tzc[i] = b2[i] + aa[i];
        }
    }
    const double ds3v = -g.s3 * g.z3 + (pass ? (sigmu - g.ds3 * g.dz3) : 0.);
    Rhs b;
    b.s = -om * it.rxs + it.tzs + it.tzc[2];
    b.ds = -om * it.rxds + 0.5 * it.tzc[0] - 0.5 * it.tzc[1];
    b.n1 = -om * it.rxn1;
    b.rhs3 = -om * it.rz3 - ds3v / g.z3;
    g.dz3 = -b.n1;
    it.bts = b.s - g.Hsd * b.ds / g.Hdd;
    it.b = b;
    const double g_dz3 = g.dz3, g_z3 = g.z3;
    PUT_BEGIN();
    PUT(gp, g, dz3);
    PUT(ip_, it, tzs);
    PUT(ip_, it, tzc);
    PUT(ip_, it, b.s);
    PUT(ip_, it, b.ds);
    PUT(ip_, it, b.n1);
    PUT(ip_, it, b.rhs3);
    PUT(ip_, it, bts);
    PUT_END();
    // ---- stages ----
    if (v.vst)
    {
        double tzv[L::NS];
        forEachCone<P>([&](auto ci) {
            constexpr int C = decltype(ci)::value;
            if (act & (1u << C))
                coneT<P, L::coneOff(C), L::coneDim(C)>(st, C, pass, om, sigmu, tzv, stz);
            else
            {
                zeroT<P, L::coneOff(C), L::coneDim(C)>(st);
#pragma unroll
                for (int i = 0; i < L::coneDim(C); i++)
                    tzv[L::coneOff(C) + i] = 0.;
            }
        });
        {
            // the LP rows of the table
            constexpr int NLP = P::NLP, LP0 = L::LP0;
            double sv[NLP], zv[NLP], rz[NLP], dsv[NLP], dzv[NLP], tz[NLP];
            ldf<NLP>(st, L::F_S + LP0, sv);
            ldf<NLP>(st, L::F_Z + LP0, zv);
            ldf<NLP>(st, L::F_RZ + LP0, rz);
            ldf<NLP>(st, L::F_DS + LP0, dsv);
            ldf<NLP>(st, L::F_DZ + LP0, dzv);
            LOADS_ISSUED();
#pragma unroll
            for (int w = 0; w < NLP; w++)
            {
                const double corr = pass ? (sigmu - dsv[w] * dzv[w]) / sv[w] : 0.;
                tz[w] = (act & (1u << (L::NCONES + w))) ? (zv[w] / sv[w]) * om * rz[w] - zv[w] + corr : 0.;
            }
            stf<NLP>(st, L::F_TZ + LP0, tz);
#pragma unroll
            for (int w = 0; w < NLP; w++)
                tzv[LP0 + w] = tz[w];
        }
        double uh[3], gw[NV], gdl;
        double rxw[NV], hdw[NV], beta[NV];
        ldf<3>(st, L::F_UHAT, uh);
        ldf<NV>(st, L::F_RXW, rxw);
        ldf<NV>(stz, L::F_HDW, hdw); // (all zero in SCvx mode: delta_k is not a variable there)
        const double rxd = st[L::F_RXD], hdd = st[L::F_HDD];
        LOADS_ISSUED();
        LTmul<P>(ip, fm, tzv, uh, gw, &gdl);
        const double bxd = -om * rxd + gdl;
        const double q = bxd / hdd;
#pragma unroll
        for (int j = 0; j < NV; j++)
        {
            rxw[j] = -om * rxw[j] + gw[j];
            beta[j] = (fm & (1u << j)) ? 0. : rxw[j] - hdw[j] * q;
        }
        st[L::F_BXD] = bxd; // (F_BXW is an input of the initialisation's kktPrep only: not stored here)
        stf<NV>(v.xs, L::X_BETA, beta);
    }
    if (v.vsg)
    {
        forSegChunks<P, chunkFor<P>(IPM_RHS_CHUNK, IPM_RHS_CHUNK_W)>([&](auto i0, auto n) { rhsSegChunk<P, decltype(i0)::value, decltype(n)::value, PASS>(sg, makeSegLds<P>(c, v.k), v.xs, om, sigmu, g_z3, g_dz3); });
    }
    WAVE_SYNC();
}

// ---- recover the eliminated variables, dz / ds, step length (pass 0: centering parameter) ----
template <class P, int I0, int N, int PASS>
__device__ inline void dirSegChunk(const SV &sg, const SegLds &sl, const SV &xs, const SV &xsz, double om, double sigmu, double z3, double dz3,
                                   double dsig, double &ainv, double &ainv_d, double &sumdnb)
{
    using L = Lay<P>;
    SegState<N> q;
    double vl[N], bcl[N], p1[N], p2[N];
    ldf<N>(xs, L::X_VL + I0, vl);
    ldf<N>(xsz, L::X_BCL + I0, bcl); // border column: zero in SCvx mode (not stored)
    ldSegState<P, I0, N>(sg, sl, q);
    if (PASS)
    {
        ldf<N>(sg, G_DS1 * L::NL + I0, p1);
        ldf<N>(sg, G_DS2 * L::NL + I0, p2);
    }
    LOADS_ISSUED();
    double o1[N], o2[N];
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        const SegRhsRow r = segRhsRow<PASS>(q.nu[i], q.nub[i], q.s1[i], q.z1[i], q.s2[i], q.z2[i], q.lam[i], PASS ? p1[i] : 0., PASS ? p2[i] : 0., om,
                                            sigmu, z3, dz3);
        const SegDirRow d = segDirRow(r, vl[i], bcl[i], om, dsig);
        sumdnb += d.dnub;
#if SEG_FAST_RCP
        double m1 = -d.ds1 * r.is1, m2 = -d.dz1 * r.iz1, m3 = -d.ds2 * r.is2, m4 = -d.dz2 * r.iz2;
#else
        double m1 = -d.ds1 / q.s1[i], m2 = -d.dz1 / q.z1[i], m3 = -d.ds2 / q.s2[i], m4 = -d.dz2 / q.z2[i];
#endif
        if (IPM_SPLIT_STEPS && PASS)
        {
            // corrector: the slacks' bound (m1, m3) and the multipliers' (m2, m4) apart
            m1 = m1 > m3 ? m1 : m3;
            m2 = m2 > m4 ? m2 : m4;
            ainv = m1 > ainv ? m1 : ainv;
            ainv_d = m2 > ainv_d ? m2 : ainv_d;
        }
        else
        {
            m1 = m1 > m2 ? m1 : m2;
            m3 = m3 > m4 ? m3 : m4;
            m1 = m1 > m3 ? m1 : m3;
            ainv = m1 > ainv ? m1 : ainv;
        }
        o1[i] = d.ds1 * d.dz1;
        o2[i] = d.ds2 * d.dz2;
    }
    if (!PASS)
    {
        // predictor: the corrector's right-hand side needs the products ds_aff dz_aff only (segRhsRow<1>).  The corrector stores
        // NOTHING: phUpdate recomputes the final direction of a row from the same inputs (state, products, vl)
        stf<N>(sg, G_DS1 * L::NL + I0, o1);
        stf<N>(sg, G_DS2 * L::NL + I0, o2);
    }
}
// One chunk as a function of its own: inside phDirSeg the scheduler of the (long) enclosing block turned the last chunk into
// load / spill / load chains with one exposed memory round trip each; compiled alone a chunk is one batch of loads, the
// arithmetic, one batch of stores.  (Calls cost nothing here: no callee-saved registers, see PHASE_FN.)
struct DirChunkOut
{
    double ainv, ainv_d, sumdnb;
};
template <class P, int I0, int N, int PASS>
PHASE_FN DirChunkOut dirSegChunkFn(const PRIV Ctx *cin, double om, double sigmu, double z3, double dz3, double dsig, double ainv, double ainv_d, double sumdnb)
{
    EMU_PHASE(PASS ? "phDirSeg<1>" : "phDirSeg<0>");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const int k = c.lane, K = c.K;
    DirChunkOut o{ainv, ainv_d, sumdnb}; // running maxima / sum of this lane (same order of additions as one pass over all rows)
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        const SV xs = makeSX(c.sx, L::XREC, K, unsigned(k));
        dirSegChunk<P, I0, N, PASS>(sg, makeSegLds<P>(c, k), xs, padView(xs, scvxMode(c.ip)), om, sigmu, z3, dz3, dsig, o.ainv, o.ainv_d, o.sumdnb);
    }
    return o;
}
// Newton direction, part 1: sigma row (border correction) and the stage cones.  Leaves dsig / ddsg and its share of the
// step-length bound (Iter::part_ainv) and of the finiteness check (Iter::part_fin) in the wave-uniform state.
template <class P, int PASS>
PHASE_FN void phDirStage(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE(PASS ? "phDirStage<1>" : "phDirStage<0>");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    constexpr int pass = PASS; // 0: predictor (affine), 1: corrector -- two instantiations, no run-time branches around loads
    const Views v = makeViews<P>(c);
    const SV &st = v.st, &sg = v.sg, &dy = v.dy;
    const double *ip = c.ip;
    const unsigned act = v.act;
    const bool scvx = scvxMode(ip);
    const SV stz = padView(v.st, scvx);
    Glob g;   // the fields this phase produces (the rest is read where it is needed)
    Iter it;
    it.bad = ip_->bad;
    g.schur = gp->schur;
    const double sigma_c = pass ? double(ip_->sigma_c) : 0.;
    const double om = 1. - sigma_c;
    // ---- sigma row: border correction (and the border's Schur complement, once per factorisation) ----
    {
        double cv = 0., bs = 0.;
        if (v.vsg)
        {
            double S[L::NL], vl[L::NL], bcl[L::NL];
            // SCvx (fixed final time): S = 0, so neither product needs its other factor -- all three through the padded view
            ldf<L::NL>(padView(dy, scvx), L::DY_S, S);
            ldf<L::NL>(padView(v.xs, scvx), L::X_VL, vl);
            ldf<L::NL>(padView(v.xs, scvx), L::X_BCL, bcl);
#pragma unroll
            for (int i = 0; i < L::NL; i++)
            {
                cv -= S[i] * vl[i];
                bs -= S[i] * bcl[i];
            }
        }
        cv = waveSumDpp(cv);
        if (pass == 0)
        {
            bs = waveSumDpp(bs);
            g.schur = gp->hsig - bs;
        }
        if (!(g.schur > 0.))
        {
            it.bad = 1;
#ifdef SCPP_HIP_EMU
            if (c.lane == 0 && getenv("SCPP_EMU_DEBUG"))
                printf("[emu] schur %g hsig %g\n", g.schur, double(gp->hsig));
#endif
        }
        g.dsig = (ip_->bts - cv) / g.schur;
        g.ddsg = (ip_->b.ds - gp->Hsd * g.dsig) / gp->Hdd;
    }
    // ainv: 1 / (largest step that keeps the slacks in their cones); ainv_d: the same for the multipliers -- kept apart in the corrector pass only
    // (IPM_SPLIT_STEPS), folded into ainv in the predictor pass, whose centring rule uses the common step length
    constexpr bool split = IPM_SPLIT_STEPS && pass != 0;
    double ainv = 0., ainv_d = 0., finite_chk = g.dsig * 0. + g.ddsg * 0.;
    if (v.vst)
    {
        double Ld[L::NS];
        {
            double dw[NV], bcw[NV], hdw[NV], uh[3];
            ldf<NV>(v.xs, L::X_VW, dw);
            ldf<NV>(padView(v.xs, scvx), L::X_BCW, bcw);
            ldf<NV>(stz, L::F_HDW, hdw);
            ldf<3>(st, L::F_UHAT, uh);
            const double bxd = st[L::F_BXD], hdd = st[L::F_HDD];
            LOADS_ISSUED(); // (without it the loads that only the SC branch below consumes sink into it, one round trip each)
            double acc = 0.;
#pragma unroll
            for (int j = 0; j < NV; j++)
            {
                dw[j] -= bcw[j] * g.dsig;
                acc += hdw[j] * dw[j];
                finite_chk += dw[j] * 0.; // NaN / Inf -> NaN
            }
            const double ddl = (ip[IP_SCVX] != 0.) ? 0. : (bxd - acc) / hdd;
            if (pass != 0) // the predictor's dw / ddelta are consumed in this phase only
            {
                stf<NV>(st, L::F_DW, dw);
                st[L::F_DDL] = ddl;
            }
            Lmul<P>(ip, act, dw, ddl, uh, Ld);
        }
        forEachCone<P>([&](auto ci) {
            constexpr int C = decltype(ci)::value;
            if (act & (1u << C))
            {
                const double a0 = coneDir<P, L::coneOff(C), L::coneDim(C)>(st, C, om, Ld, pass != 0, stz, ainv_d);
                ainv = a0 > ainv ? a0 : ainv;
            }
        });
        {
            constexpr int NLP = P::NLP, LP0 = L::LP0;
            double sv[NLP], zv[NLP], rz[NLP], tz[NLP], dzv[NLP], dsv[NLP];
            ldf<NLP>(st, L::F_S + LP0, sv);
            ldf<NLP>(st, L::F_Z + LP0, zv);
            ldf<NLP>(st, L::F_RZ + LP0, rz);
            ldf<NLP>(st, L::F_TZ + LP0, tz);
            LOADS_ISSUED();
#pragma unroll
            for (int w = 0; w < NLP; w++)
            {
                const bool on = act & (1u << (L::NCONES + w));
                dzv[w] = on ? -(zv[w] / sv[w]) * Ld[LP0 + w] + tz[w] : 0.;
                dsv[w] = on ? -om * rz[w] + Ld[LP0 + w] : 0.;
                const double a1 = on ? -dsv[w] / sv[w] : 0., a2 = on ? -dzv[w] / zv[w] : 0.;
                ainv = a1 > ainv ? a1 : ainv;
                ainv_d = a2 > ainv_d ? a2 : ainv_d;
            }
            stf<NLP>(st, L::F_DZ + LP0, dzv);
            stf<NLP>(st, L::F_DS + LP0, dsv);
        }
    }
    if (!split)
        ainv = ainv_d > ainv ? ainv_d : ainv;
    ainv = waveMaxDpp(ainv);
    if (split)
        ainv_d = waveMaxDpp(ainv_d);
    finite_chk = waveSumDpp(finite_chk);
    it.part_ainv = ainv;
    it.part_ainv_d = ainv_d;
    it.part_fin = finite_chk;
    PUT_BEGIN();
    if (pass == 0)
        PUT(gp, g, schur);
    PUT(ip_, it, bad);
    PUT(ip_, it, part_ainv);
    if (split)
        PUT(ip_, it, part_ainv_d);
    PUT(ip_, it, part_fin);
    PUT(gp, g, dsig);
    PUT(gp, g, ddsg);
    PUT_END();
}
// part 2: the segment rows (nu, nu_b and their LP cones), the wave-uniform rows and the step length
template <class P, int PASS>
PHASE_FN void phDirSeg(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE(PASS ? "phDirSeg<1>" : "phDirSeg<0>");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    constexpr int pass = PASS; // 0: predictor (affine), 1: corrector -- two instantiations, no run-time branches around loads
    const Views v = makeViews<P>(c);
    const SV &sg = v.sg;
    Glob g; // the fields this phase produces
    Iter it;
    it.bad = ip_->bad;
    g.dsig = gp->dsig;
    g.ddsg = gp->ddsg;
    const double sigma_c = pass ? double(ip_->sigma_c) : 0.;
    const double om = 1. - sigma_c;
    constexpr bool split = IPM_SPLIT_STEPS && pass != 0; // corrector: primal (ainv) and dual (ainv_d) step-length bounds apart
    double ainv = ip_->part_ainv, ainv_d = split ? double(ip_->part_ainv_d) : 0., finite_chk = ip_->part_fin;
    double sumdnb = 0.;
    const double sigmu = sigma_c * double(ip_->mu), z3 = gp->z3, dz3 = gp->dz3;
    forSegChunks<P, chunkFor<P>(IPM_DIR_CHUNK, IPM_DIR_CHUNK_W)>([&](auto i0, auto n) {
        const DirChunkOut o = dirSegChunkFn<P, decltype(i0)::value, decltype(n)::value, PASS>(cin, om, sigmu, z3, dz3, g.dsig, ainv, ainv_d, sumdnb);
        ainv = o.ainv;
        ainv_d = o.ainv_d;
        sumdnb = o.sumdnb;
    });
    sumdnb = waveSumDpp(sumdnb);
    const Glob gt = reloadPriv(gp);
    const Iter itt = reloadPriv(ip_);
    // a non-finite Newton direction (breakdown of the factorisation near the end of the path) must not be applied
    // (sumdnb carries dlam through dnu / dnub)
    finite_chk = finite_chk + waveSumDpp(sumdnb * 0.);
    if (!(finite_chk == 0.))
        it.bad = 1;
    g.dn1 = sumdnb - (gt.s3 / gt.z3) * gt.dz3 - itt.b.rhs3;
    g.dzs = -(gt.zs / gt.ss) * g.dsig + itt.tzs;
    g.dss = -om * itt.rzs + g.dsig;
    {
        const double m1 = -g.dss / gt.ss, m2 = -g.dzs / gt.zs;
        ainv = m1 > ainv ? m1 : ainv;
        ainv_d = m2 > ainv_d ? m2 : ainv_d;
    }
    g.ds3 = -om * itt.rz3 + (g.dn1 - sumdnb);
    {
        const double m1 = -g.ds3 / gt.s3, m2 = -gt.dz3 / gt.z3;
        ainv = m1 > ainv ? m1 : ainv;
        ainv_d = m2 > ainv_d ? m2 : ainv_d;
    }
    {
        const double Ld[3] = {0.5 * g.ddsg, -0.5 * g.ddsg, g.dsig};
        double aa[3];
        cone::applyWinv2(gt.seta, gt.sw, 3, Ld, aa);
        for (int i = 0; i < 3; i++)
        {
            g.dzc3[i] = -aa[i] + itt.tzc[i];
            g.dsc3[i] = -om * itt.rzc[i] + Ld[i];
        }
        cone::applyWinv(gt.seta, gt.sw, 3, g.dsc3, g.dsC);
        cone::applyW(gt.seta, gt.sw, 3, g.dzc3, g.dzC);
        const double a1 = cone::stepInv(3, gt.lamC, g.dsC), a2 = cone::stepInv(3, gt.lamC, g.dzC);
        ainv = a1 > ainv ? a1 : ainv;
        ainv_d = a2 > ainv_d ? a2 : ainv_d;
    }
    if (!split)
        ainv = ainv_d > ainv ? ainv_d : ainv;
    ainv = waveMaxDpp(ainv);
    if (split)
    {
        ainv_d = waveMaxDpp(ainv_d);
        if (itt.common_step) // the repeat of a failed attempt: one step length, the smaller of the two (ECOS's rule)
            ainv = ainv_d = (ainv_d > ainv ? ainv_d : ainv);
    }
    if (pass == 0)
    {
        double alpha_a = ainv > 0. ? 1. / ainv : 1.;
        alpha_a = alpha_a < 1. ? alpha_a : 1.;
        double sc = (1. - alpha_a) * (1. - alpha_a) * (1. - alpha_a);
        sc = sc < 1e-4 ? 1e-4 : sc;
        sc = sc > 1. ? 1. : sc;
        it.sigma_c = sc;
    }
    else
    {
        double alpha = ainv > 0. ? itt.gamma / ainv : 1.;
        alpha = alpha < 1. ? alpha : 1.;
        alpha = alpha < 0.999 ? alpha : 0.999;
        alpha = alpha > 1e-8 ? alpha : 1e-8;
        it.alpha = alpha;
        double alpha_d = alpha; // ECOS: one step length
        if (split)
        {
            alpha_d = ainv_d > 0. ? itt.gamma / ainv_d : 1.;
            alpha_d = alpha_d < 1. ? alpha_d : 1.;
            alpha_d = alpha_d < 0.999 ? alpha_d : 0.999;
            alpha_d = alpha_d > 1e-8 ? alpha_d : 1e-8;
        }
        it.alpha_d = alpha_d;
    }
    PUT_BEGIN();
    if (pass == 0)
        PUT(ip_, it, sigma_c);
    else
    {
        PUT(ip_, it, alpha);
        PUT(ip_, it, alpha_d);
    }
    PUT(ip_, it, bad);
    PUT(gp, g, dn1);
    PUT(gp, g, dzs);
    PUT(gp, g, dss);
    PUT(gp, g, ds3);
    PUT(gp, g, dzc3);
    PUT(gp, g, dsc3);
    PUT(gp, g, dsC);
    PUT(gp, g, dzC);
    PUT_END();
}

// ---- x += alpha dx ; s += alpha ds ; z += alpha dz ----
template <int N>
__device__ inline void axpyFields(const SV &rec, int fDst, int fSrc, double alpha)
{
    double d[N], x[N];
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        d[i] = rec[fDst + i];
        x[i] = rec[fSrc + i];
    }
    LOADS_ISSUED();
#pragma unroll
    for (int i = 0; i < N; i++)
        rec[fDst + i] = d[i] + alpha * x[i];
}
// dst += alpha src for a stage vector that starts at the trust-region cone (rows 1 .. NP: structural zeros in SCvx mode)
template <int N, int NP>
__device__ inline void axpyPad(const SV &rec, const SV &recz, int fDst, int fSrc, double alpha)
{
    double d[N], x[N];
    ldPad<N, NP>(rec, recz, fDst, d);
    ldPad<N, NP>(rec, recz, fSrc, x);
    LOADS_ISSUED();
#pragma unroll
    for (int i = 0; i < N; i++)
        d[i] = d[i] + alpha * x[i];
    stPad<N, NP>(rec, recz, fDst, d);
}
// NF fields of N rows each in ONE load group / store group (dst field f at fDst[f], its direction at fSrc[f])
template <int N, int NF>
__device__ inline void axpyFieldGroup(const SV &rec, const int (&fDst)[NF], const int (&fSrc)[NF], double alpha)
{
    double d[NF][N], x[NF][N];
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            d[f][i] = rec[fDst[f] + i];
            x[f][i] = rec[fSrc[f] + i];
        }
    LOADS_ISSUED();
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int i = 0; i < N; i++)
            rec[fDst[f] + i] = d[f][i] + alpha * x[f][i];
}
template <class P, int I0, int N>
__device__ inline void updSegChunk(const SV &sg, const SegLds &sl, const SV &xs, const SV &xsz, double om, double sigmu, double z3, double dz3, double dsig,
                                   double alpha, double alpha_d)
{
    using L = Lay<P>;
    SegState<N> q;
    double vl[N], bcl[N], p1[N], p2[N];
    ldf<N>(xs, L::X_VL + I0, vl);
    ldf<N>(xsz, L::X_BCL + I0, bcl);
    ldSegState<P, I0, N>(sg, sl, q);
    ldf<N>(sg, G_DS1 * L::NL + I0, p1);
    ldf<N>(sg, G_DS2 * L::NL + I0, p2);
    LOADS_ISSUED();
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        const SegRhsRow r = segRhsRow<1>(q.nu[i], q.nub[i], q.s1[i], q.z1[i], q.s2[i], q.z2[i], q.lam[i], p1[i], p2[i], om, sigmu, z3, dz3);
        const SegDirRow d = segDirRow(r, vl[i], bcl[i], om, dsig);
        q.nu[i] = q.nu[i] + alpha * d.dnu;
        q.nub[i] = q.nub[i] + alpha * d.dnub;
        q.lam[i] = q.lam[i] + alpha_d * d.dlam;
        q.s1[i] = q.s1[i] + alpha * d.ds1;
        q.z1[i] = q.z1[i] + alpha_d * d.dz1;
        q.s2[i] = q.s2[i] + alpha * d.ds2;
        q.z2[i] = q.z2[i] + alpha_d * d.dz2;
    }
    stl<P, SL_NU, I0, N>(sl, q.nu);
    stl<P, SL_NUB, I0, N>(sl, q.nub);
    stl<P, SL_LAM, I0, N>(sl, q.lam);
    stf<N>(sg, G_S1 * L::NL + I0, q.s1);
    stf<N>(sg, G_Z1 * L::NL + I0, q.z1);
    stf<N>(sg, G_S2 * L::NL + I0, q.s2);
    stf<N>(sg, G_Z2 * L::NL + I0, q.z2);
}
// the LDS-resident segment fields (ipm_kernel.h: SegLdsSlot): workspace -> LDS after the initialisation, LDS -> workspace when the
// solve ends (warm start of the next launch; nothing else reads them)
template <class P, bool TO_LDS>
PHASE_FN void phSegLdsCopy(const PRIV Ctx *cin)
{
    EMU_PHASE("phSegLdsCopy");
    using L = Lay<P>;
    if constexpr (!SegInLds<P>::value)
        return; // the fields never leave the workspace
    const Ctx c = uniformCtx(cin);
    const int k = c.lane;
    if (k < c.K - 1)
    {
        const SV sg = makeSV(c.sg, (G_NFIELDS * L::NL), unsigned(k), c.pitch);
        const SegLds sl = makeSegLds<P>(c, k);
        double nu[L::NL], nub[L::NL], lam[L::NL];
        if (TO_LDS)
        {
            ldf<L::NL>(sg, G_NU * L::NL, nu);
            ldf<L::NL>(sg, G_NUB * L::NL, nub);
            ldf<L::NL>(sg, G_LAM * L::NL, lam);
            LOADS_ISSUED();
            stl<P, SL_NU, 0, L::NL>(sl, nu);
            stl<P, SL_NUB, 0, L::NL>(sl, nub);
            stl<P, SL_LAM, 0, L::NL>(sl, lam);
        }
        else
        {
            ldl<P, SL_NU, 0, L::NL>(sl, nu);
            ldl<P, SL_NUB, 0, L::NL>(sl, nub);
            ldl<P, SL_LAM, 0, L::NL>(sl, lam);
            stf<L::NL>(sg, G_NU * L::NL, nu);
            stf<L::NL>(sg, G_NUB * L::NL, nub);
            stf<L::NL>(sg, G_LAM * L::NL, lam);
        }
    }
    WAVE_SYNC();
}
template <class P>
PHASE_FN void phUpdate(const PRIV Ctx *cin, PRIV Glob *gp, PRIV Iter *ip_)
{
    EMU_PHASE("phUpdate");
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const Views v = makeViews<P>(c);
    const SV &st = v.st, &sg = v.sg;
    Glob g = loadPriv(gp);
    const double alpha = ip_->alpha, alpha_d = ip_->alpha_d;
    const bool scvx = scvxMode(c.ip);
    const SV stz = padView(v.st, scvx);
    // few, large load groups: a group's loads wait for the stores of the group before it (one memory round trip per group)
    if (v.vst)
    {
        double d[NV], x[NV];
#pragma unroll
        for (int j = 0; j < NV; j++)
        {
            d[j] = st[L::F_W + j];
            x[j] = st[L::F_DW + j];
        }
        const double dl = st[L::F_DL], ddl = st[L::F_DDL];
        LOADS_ISSUED();
#pragma unroll
        for (int j = 0; j < NV; j++)
            st[L::F_W + j] = d[j] + ((v.fm & (1u << j)) ? 0. : alpha * x[j]);
        st[L::F_DL] = dl + alpha * ddl;
        axpyPad<L::NS, P::NXV>(st, stz, L::F_S, L::F_DS, alpha);
        axpyPad<L::NS, P::NXV>(st, stz, L::F_Z, L::F_DZ, alpha_d);
    }
    if (v.vsg)
    {
        // the corrector direction of the segment rows is recomputed here from what phDirSeg<1> read (state, predictor products, the
        // block solve's multiplier direction) instead of being stored by it and loaded back: segRhsRow<1> / segDirRow
        const double sigma_c = ip_->sigma_c, om = 1. - sigma_c, sigmu = sigma_c * double(ip_->mu);
        const SV xsz = padView(v.xs, scvx);
        forSegChunks<P, chunkFor<P>(IPM_UPD_CHUNK, IPM_UPD_CHUNK_W)>([&](auto i0, auto n) {
            updSegChunk<P, decltype(i0)::value, decltype(n)::value>(sg, makeSegLds<P>(c, v.k), v.xs, xsz, om, sigmu, g.z3, g.dz3, g.dsig, alpha, alpha_d);
        });
    }
    g.sig += alpha * g.dsig;
    g.dsg += alpha * g.ddsg;
    g.n1 += alpha * g.dn1;
    g.ss += alpha * g.dss;
    g.zs += alpha_d * g.dzs;
    g.s3 += alpha * g.ds3;
    g.z3 += alpha_d * g.dz3;
    for (int i = 0; i < 3; i++)
    {
        g.sc3[i] += alpha * g.dsc3[i];
        g.zc3[i] += alpha_d * g.dzc3[i];
    }
    PUT_BEGIN();
    PUT(gp, g, sig);
    PUT(gp, g, dsg);
    PUT(gp, g, n1);
    PUT(gp, g, ss);
    PUT(gp, g, zs);
    PUT(gp, g, s3);
    PUT(gp, g, z3);
    PUT(gp, g, sc3);
    PUT(gp, g, zc3);
    PUT_END();
}

// `late`: where the tail re-reads the KernelArgs from -- the kernel-argument segment (constant memory).  The builtin that returns that
// segment's address is only valid in the KERNEL function (in a callee the compiler folds it to null): kernels pass kernelArgsLate() down.
#ifdef SCPP_HIP_EMU
typedef const KernelArgs ConstKernelArgs;
#define KERNEL_TAIL_ARGS(t, a, late) const KernelArgs &t = a
#else
typedef const __attribute__((address_space(4))) KernelArgs ConstKernelArgs;
__device__ inline ConstKernelArgs *kernelArgsLate()
{
    ConstKernelArgs *p = (ConstKernelArgs *)__builtin_amdgcn_kernarg_segment_ptr(); // explicit arguments start at offset 0
    asm volatile("" : "+s"(p));                                                      // opaque: the loads stay where they are used
    return p;
}
__device__ inline ConstKernelArgs *opaqueArgs(ConstKernelArgs *p)
{
    asm volatile("" : "+s"(p));
    return p;
}
#define KERNEL_TAIL_ARGS(t, a, late) ConstKernelArgs &t = *opaqueArgs(late)
#endif
#ifndef IPM_WAVES_PER_SIMD
#define IPM_WAVES_PER_SIMD 2
#endif
#ifdef IPM_PROFILE
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, t0, t1) prof[slot] += double((t1) - (t0))
#else
#define PROF_T(var)
#define PROF_ADD(slot, t0, t1)
#endif
// One whole sub-problem solve of instance `inst` by the calling wavefront: the body of ipm_kernel and of the solve step of the persistent
// SCvx kernel (scvx_persistent.h).  The tail re-reads its arguments from the kernel-argument segment (KERNEL_TAIL_ARGS): every kernel that
// calls this takes its KernelArgs as the FIRST kernel parameter.
template <class P>
__device__ __forceinline__ void ipmSolveInstance(const KernelArgs &a, const int inst, ConstKernelArgs *late)
{
    using L = Lay<P>;
    constexpr int NX = P::NX, NU = P::NU;
#ifdef IPM_PROFILE
    double prof[12] = {0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.};
    const long long t_kernel0 = clock64();
#endif
    if (inst >= a.B)
        return;
    if (a.active && a.active[inst] == 0)
        return;
    __shared__ TileShared sh;
#ifdef IPM_PROFILE
    if (threadIdx.x == 0)
        for (int i = 0; i < 10; i++)
            sh.prof[i] = 0.;
#endif
    const int K = a.K, lane = threadIdx.x, k = lane;
    Ctx c;
    c.K = K;
    c.lane = lane;
    double *ws = a.ws + size_t(inst) * workspaceDoubles<P>(K);
    c.sx = ws;
    c.st = ws + size_t(K) * L::XREC;
    c.pitch = recPitch(K);
    c.sg = c.st + size_t(c.pitch) * L::STREC;
    c.dy = c.sg + size_t(c.pitch) * (G_NFIELDS * L::NL);
    c.fac = ws + facOffset<P>(K); // on a 128-byte line (ipm_kernel.h: FACREC)
    c.sv = c.fac + size_t(K) * L::FACREC;
    c.gsave = c.sv + size_t(K) * SVREC;
    c.A = a.A + size_t(inst) * (K - 1) * NX * NX;
    c.B = a.Bm + size_t(inst) * (K - 1) * NX * NU;
    c.C = a.C + size_t(inst) * (K - 1) * NX * NU;
    c.S = a.S + size_t(inst) * (K - 1) * NX;
    c.Z = a.Z + size_t(inst) * (K - 1) * NX;
    c.ip = a.ip + size_t(inst) * IP_N;
    EMU_PHASE("solve: outside the phases");
    EMU_TRAFFIC_REGION("exchange [K][XREC]", c.sx, size_t(K) * L::XREC * 8, 8, size_t(L::XREC) * 8);
    EMU_TRAFFIC_REGION("stage [STREC][K]", c.st, size_t(c.pitch) * L::STREC * 8, size_t(c.pitch) * 8, 0);
    EMU_TRAFFIC_REGION("segment [G_NFIELDS*NL][K]", c.sg, size_t(c.pitch) * (G_NFIELDS * L::NL) * 8, size_t(c.pitch) * 8, 0);
    EMU_TRAFFIC_REGION("dd copy [DYNREC][K]", c.dy, size_t(c.pitch) * L::DYNREC * 8, size_t(c.pitch) * 8, 0);
    EMU_TRAFFIC_REGION("factor [K][FACREC]", c.fac, size_t(K) * L::FACREC * 8, 8, size_t(L::FACREC) * 8);
    EMU_TRAFFIC_REGION("saved columns [K][SVREC]", c.sv, size_t(K) * SVREC * 8, 8, size_t(SVREC) * 8);
    EMU_TRAFFIC_REGION("dd A [K-1][NX][NX]", c.A, size_t(K - 1) * NX * NX * 8, 8, size_t(NX) * NX * 8);
    EMU_TRAFFIC_REGION("dd B [K-1][NX][NU]", c.B, size_t(K - 1) * NX * NU * 8, 8, size_t(NX) * NU * 8);
    EMU_TRAFFIC_REGION("dd C [K-1][NX][NU]", c.C, size_t(K - 1) * NX * NU * 8, 8, size_t(NX) * NU * 8);
    // dynamic LDS (launchIpm: segLdsBytes<P>(K) [+ pad]): the LDS-resident segment fields
#ifdef SCPP_HIP_EMU
    static double seg_lds[NSEGLDS * 16 * 64];
    c.segl = seg_lds;
#else
    extern __shared__ __attribute__((aligned(16))) double seg_lds[]; // 16-byte base: DiscLds members are aligned(16) (b128 LDS accesses), whatever the static LDS before it adds up to
    c.segl = (LDSP double *)seg_lds;
#endif
    const Settings opt = a.opt;

    // wave-uniform state in LDS: one copy per wavefront, handed to the out-of-line phases by LDS address
    __shared__ Glob g;
    __shared__ Iter it;
    __shared__ Ctx cshared;
    PRIV Glob *gp = (PRIV Glob *)&g;
    PRIV Iter *itp = (PRIV Iter *)&it;
    const PRIV Ctx *cs = (const PRIV Ctx *)&cshared;
    if (lane == 0)
        cshared = c;
    WAVE_SYNC();
    const double wtrx = a.wtrx[inst];
    it.wtrx = wtrx;
    it.w_t = c.ip[IP_WT];
    it.w_trt = c.ip[IP_WTRT];
    it.w_vc = c.ip[IP_WVC];
    it.sigbar = a.sigma[inst];
    it.gamma = opt.gamma;
    it.sigma_c = 0.;
    it.alpha = 1.;
    it.alpha_d = 1.;
    it.bk_valid = 0;
    it.bk_sig = it.bk_dsg = it.bk_n1 = it.pres_prev = 0.;

    if (a.Xold && k < K)
    {
        EMU_TRAFFIC_MANUAL("td X, U -> old_td snapshot (plain pointers)", NX + NU, false);
        EMU_TRAFFIC_MANUAL("td X, U -> old_td snapshot (plain pointers)", NX + NU, true);
        const size_t o = size_t(inst) * K + k;
        for (int j = 0; j < NX; j++)
            a.Xold[o * NX + j] = a.X[o * NX + j];
        for (int j = 0; j < NU; j++)
            a.Uold[o * NU + j] = a.U[o * NU + j];
    }
    PROF_T(tp0);
    int warm = (a.warm && a.warm[inst] != 0) ? 1 : 0;
    // a re-solve on unchanged data (only where the previous solve of this workspace ran to its end on them: it is warm-startable)
    const int dd_same = (a.dd_fresh && a.dd_fresh[inst] == 0 && warm) ? 1 : 0;
    int status = -1, iter = 0, iter_total = 0;
    bool use_backup = false;
    // a warm start that breaks down is repeated from ECOS's cold initialisation; a cold attempt that fails with primal and dual step lengths of their
    // own (IPM_SPLIT_STEPS) is repeated with ECOS's common step length -- round 6: ONE of 1 048 576 soak trajectories (instance 454733, its first
    // solve) failed under the split rule: at its 19th iteration, with the gap already at 9e-8 relative, a rounding-level loss in the step left the
    // primal residual at 1.8e-8 against the 1e-8 tolerance, two iterations later the dual residual was at 1e-4 and the factorisation broke down (the
    // emulator and the twin, equal to the device to rounding, converge at that very iteration; the common-step path was through after 18).  What a
    // failed attempt costs is its iterations; a solve that succeeds is untouched.  profiles/r06_soak_failure_instance_454733.json, tools/r06_trace_failure.py
    // ... and a WARM-STARTED solve of SCAlgorithm's sub-problem takes the common step length from the start: those solves sit next to the fixed point the SC
    // iteration stalls at (5 iterations each at best), and there the split rule was measured worse on the GPU -- 4096 closed loops of SC_sim's shape, same box,
    // alternating: 114.7 against 111.7 iterations and 0.390 against 0.361 s per 15-iteration solve (profiles/r06_ab_scsim_split_steps.json); cold SC solves
    // (127.6 against 132.9 iterations) and every SCvx solve (333.6 against 367.9 per trajectory) keep the split rule
    int common_step = (IPM_SPLIT_STEPS && warm && c.ip[IP_SCVX] == 0.) ? 1 : 0;
    for (int attempt = 0; attempt < 3; attempt++)
    {
    it.common_step = common_step;
    phSetup<P>(cs, a.X + size_t(inst) * K * NX, a.U + size_t(inst) * K * NU, a.uhat + size_t(inst) * K * 3, gp, itp, warm, dd_same);
    if (warm)
    {
        // sub-problems of consecutive SC iterations are close: restart from the previous primal-dual point
        phWarmInit<P>(cs, gp, itp);
    }
    else
    {
        // =============== initialisation (ECOS init, W = I) ===============
        phInitPrimalRhs<P>(cs, gp, itp);
        {
            const RhsSpec sp = specBorderPlus();
            factorSweepAny<P>(cs, sh, sp);
            bwdSweepAny<P>(cs, sp);
        }
        phInitPrimalFinish<P>(cs, gp, itp);
        phInitDualRhs<P>(cs, gp, itp);
        {
            const RhsSpec sp = specSingle();
            fwdSweepAny<P>(cs, sp);
            bwdSweepAny<P>(cs, sp);
        }
        phInitDualFinish<P>(cs, gp, itp);
    }
    phDataNorms<P>(cs, gp, itp, dd_same);
    phSegLdsCopy<P, true>(cs);
    PROF_T(tp1);
    PROF_ADD(0, tp0, tp1);

    status = -1;
    iter = 0;
    use_backup = false;
    it.bk_valid = 0;
    it.bad = 0;
    bool inacc_ok = false, bk_prev = false;
    double pres_prev = 0.;
    for (iter = 0;; iter++)
    {
        PROF_T(tr0);
        // (the step of the previous iteration is applied on the way into this pass: phResiduals<P, true>)
#if IPM_FUSE_UPDATE
        if (iter == 0)
            phResiduals<P, false>(cs, gp, itp);
        else
            phResiduals<P, true>(cs, gp, itp);
#else
        phResiduals<P, false>(cs, gp, itp);
#endif
        PROF_T(tr1);
        PROF_ADD(1, tr0, tr1);
        {
            const double pres = it.pres, dres = it.dres, gap = it.gap;
            const double apc = fabs(it.pcost) > 1e-300 ? fabs(it.pcost) : 1e-300;
            const double relgap = gap / apc;
#ifdef SCPP_HIP_EMU
            if (c.lane == 0 && getenv("SCPP_EMU_DEBUG"))
                printf("[emu] iter %d pres %.3e dres %.3e gap %.3e pcost %.9e | last step: alpha %.3e alpha_d %.3e sigma_c %.3e\n", iter, pres, dres, gap, double(it.pcost),
                       double(it.alpha), double(it.alpha_d), double(it.sigma_c));
#endif
            const bool nonfinite = !(pres == pres) || !(dres == dres) || !(gap == gap) || fabs(pres) > 1e300 || fabs(dres) > 1e300 || fabs(gap) > 1e300 ||
                                   fabs(it.pcost) > IPM_BLOWN || ipmGapBroken(gap); // a BLOWN-UP iterate is a broken one: see IPM_BLOWN, IPM_NEG_GAP
            // ECOS-style safeguarding: residual explosion after an acceptable iterate -> return that iterate
            if (nonfinite || (bk_prev && iter > 0 && (pres > 500. * pres_prev || gap < 0.)))
            {
                status = bk_prev ? 0 : -2;
                use_backup = bk_prev;
                break;
            }
            pres_prev = pres;
            bk_prev = it.bk_valid != 0;
            if (pres < opt.feastol && dres < opt.feastol && (gap < opt.abstol || relgap < opt.reltol))
            {
                status = 0;
                break;
            }
            // ECOS's reduced-accuracy exit (feastol_inacc 1e-4, abstol_inacc / reltol_inacc 5e-5): an iteration limit or a
            // numerical breakdown at an iterate that already meets the relaxed tolerances returns that iterate
            inacc_ok = pres < 1e-4 && dres < 1e-4 && (gap < 5e-5 || relgap < 5e-5);
            if (iter >= opt.maxit)
            {
                status = inacc_ok ? 0 : -1;
                break;
            }
        }
        phScalings<P>(cs, gp, itp);
        PROF_T(tr2);
        PROF_ADD(2, tr1, tr2);
        if (it.bad)
        {
            status = inacc_ok ? 0 : -2;
            break;
        }
        for (int pass = 0; pass < 2; pass++)
        {
            PROF_T(tq0);
            if (pass == 0)
                phRhs<P, 0>(cs, gp, itp);
            else
                phRhs<P, 1>(cs, gp, itp);
            PROF_T(tq1);
            PROF_ADD(4, tq0, tq1);
            if (pass == 0)
            {
                // one factorisation per iteration, fused with the forward substitution of the sigma border
                // column and of the affine right-hand side.  SCvx (fixed final time): S = 0, the border column is identically
                // zero -- it is neither solved for nor stored; its readers see zeros through an out-of-range view (padView)
                const RhsSpec sp = (c.ip[IP_SCVX] != 0.) ? specSingle() : specBorderPlus();
                factorSweepAny<P>(cs, sh, sp);
                PROF_T(tf1);
                PROF_ADD(3, tq1, tf1);
                bwdSweepAny<P>(cs, sp);
                PROF_T(tf2);
                PROF_ADD(9, tf1, tf2);
            }
            else
            {
                const RhsSpec sp = specSingle();
                fwdSweepAny<P>(cs, sp);
                PROF_T(tf1);
                PROF_ADD(10, tq1, tf1);
                bwdSweepAny<P>(cs, sp);
                PROF_T(tf2);
                PROF_ADD(9, tf1, tf2);
            }
            PROF_T(tq2);
            PROF_ADD(5, tq1, tq2);
            if (pass == 0)
            {
                phDirStage<P, 0>(cs, gp, itp);
                phDirSeg<P, 0>(cs, gp, itp);
            }
            else
            {
                phDirStage<P, 1>(cs, gp, itp);
                phDirSeg<P, 1>(cs, gp, itp);
            }
            PROF_T(tq3);
            PROF_ADD(6, tq2, tq3);
            if (it.bad)
                break;
        }
        if (it.bad)
        {
            status = inacc_ok ? 0 : -2;
            break;
        }
#if !IPM_FUSE_UPDATE
        phUpdate<P>(cs, gp, itp);
#endif
    }

    phSegLdsCopy<P, false>(cs);
    iter_total += iter;
    if (status == 0)
        break;
    if (warm)
    {
        warm = 0; // repeat from the cold initialisation (with the cold attempt's own rule: split step lengths first)
        common_step = 0;
    }
    else if (IPM_SPLIT_STEPS && !common_step)
        common_step = 1; // repeat (cold) with the common step length
    else
        break;
    }
    iter = iter_total;
    // =============== outputs: readSolution + SC bookkeeping ===============
    // The output pointers are re-read from the kernel-argument segment here instead of being carried (as spilled SGPRs) across
    // the whole solve: nothing below the main loop keeps a kernel argument alive above it.
    KERNEL_TAIL_ARGS(t, a, late);
    const bool vst = k < K;
    const SV st = makeSV(c.st, L::STREC, unsigned(vst ? k : 0), c.pitch);
    // W / delta to report: the current iterate, or the restored best one (use_backup)
    const int fW = use_backup ? int(L::F_WBK) : int(L::F_W);
    double sum_delta = 0.;
    if (vst)
        sum_delta = st[fW + 16];
    sum_delta = wave_sum(sum_delta);
    const double n1 = use_backup ? it.bk_n1 : g.n1, sig = use_backup ? it.bk_sig : g.sig, dsg = use_backup ? it.bk_dsg : g.dsg;
#ifdef IPM_PROFILE
    if (t.dbg && lane == 0)
    {
        double *d = t.dbg + size_t(inst) * 32;
        prof[11] = double(clock64() - t_kernel0);
        for (int i = 0; i < 12; i++)
            d[8 + i] = prof[i];
        for (int i = 0; i < 10; i++)
            d[20 + i] = sh.prof[i];
    }
#endif
    if (t.dbg && lane == 0)
    {
        double *d = t.dbg + size_t(inst) * 32;
        d[0] = it.pcost;
        d[1] = it.gap;
        d[2] = it.pres;
        d[3] = it.dres;
        d[4] = iter;
        d[5] = status;
        d[6] = n1;
        d[7] = sum_delta;
    }
    if (status == 0)
    {
        if (vst)
        {
            EMU_PHASE("solve: outputs");
            EMU_TRAFFIC_MANUAL("td X, U written (plain pointers)", NX + NU, true);
            double *Xo = t.X + (size_t(inst) * K + k) * NX, *Uo = t.U + (size_t(inst) * K + k) * NU;
            // states / inputs the table pins for the whole horizon are written as their constant (0)
#pragma unroll
            for (int i = 0; i < NX; i++)
                Xo[i] = L::XINV.v[i] >= 0 ? double(st[fW + (L::XINV.v[i] >= 0 ? L::XINV.v[i] : 0)]) : 0.;
#pragma unroll
            for (int i = 0; i < NU; i++)
                Uo[i] = L::UINV.v[i] >= 0 ? double(st[fW + (L::UINV.v[i] >= 0 ? L::UINV.v[i] : 0)]) : 0.;
        }
        if (lane == 0 && c.ip[IP_SCVX] == 0. && c.ip[IP_FIXEDT] == 0.)
            t.sigma[inst] = sig; // SCvx / SC with free_final_time false: fixed final time (the sigma block is a decoupled dummy)
    }
    if (lane == 0)
    {
        // wave-uniform part of the final primal-dual point (the rest is in the records) for a warm start of the next solve
        double *gs = c.gsave;
        gs[0] = g.sig;
        gs[1] = g.dsg;
        gs[2] = g.n1;
        gs[3] = g.ss;
        gs[4] = g.zs;
        gs[5] = g.s3;
        gs[6] = g.z3;
        for (int i = 0; i < 3; i++)
        {
            gs[7 + i] = g.sc3[i];
            gs[10 + i] = g.zc3[i];
        }
        if (t.warm)
            t.warm[inst] = (status == 0 && !use_backup) ? 1 : 0;
        t.ipm_iters[inst] += iter;
        t.norm1_nu[inst] = n1;
        t.sum_delta[inst] = sum_delta;
        t.delta_sigma[inst] = dsg;
        if (t.do_sc_update)
        {
            t.sc_iters[inst] += 1;
            if (status != 0)
            {
                t.status[inst] = status; // solver failure: reference would std::terminate (SCAlgorithm.cpp:94-98)
                t.active[inst] = 0;
            }
            else
            {
                if (n1 < t.nu_tol)
                    t.wtrx[inst] = wtrx * 2.;
                const int conv = (sum_delta < t.delta_tol && n1 < t.nu_tol) ? 1 : 0;
                if (conv)
                {
                    t.converged[inst] = 1;
                    t.active[inst] = 0;
                }
                else if (t.sc_iters[inst] >= t.max_sc_iterations)
                    t.active[inst] = 0;
            }
        }
        else
            t.status[inst] = status;
    }
}
template <class P>
__global__ void __launch_bounds__(WAVE, IPM_WAVES_PER_SIMD) __attribute__((disable_tail_calls)) ipm_kernel(KernelArgs a)
{
#ifdef SCPP_HIP_EMU
    ipmSolveInstance<P>(a, blockIdx.x, &a);
#else
    ipmSolveInstance<P>(a, blockIdx.x, kernelArgsLate());
#endif
}

} // namespace ipm
} // namespace scpp
