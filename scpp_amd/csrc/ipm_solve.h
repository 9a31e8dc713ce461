// Main loop of the batched interior-point kernel (one wavefront per instance).
// Line-by-line device counterpart of oracle/structured_ipm.hpp::RQStructuredSocp::run (the scalar
// twin carries the derivation); see ipm_kernel.h for layout and tile helpers.
#pragma once
#include "ipm_kernel.h"
#include "sweeps.h"

namespace scpp
{
namespace ipm
{

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
    int do_sc_update;            // 1: apply readSolution + convergence logic ; 0: plain sub-problem solve
    Settings opt;
    double *dbg;                 // optional [B][8]: pcost, gap, pres, dres, iters, status
};

struct Rhs
{
    double s, ds, n1, rhs3;
};

__device__ inline double dLP(bool identity, double sv, double zv) { return identity ? 1. : zv / sv; }

// per-lane preparation of the factorisation: segment scalars, stage Hessian, sigma block
__device__ inline void prepareFactor(const Ctx &c, bool identity, Glob &g)
{
    const int k = c.lane, K = c.K;
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, SEGREC, unsigned(k));
        for (int i = 0; i < NL; i++)
        {
            if (identity)
            {
                sg[G_EINV * NL + i] = 0.5;
                sg[G_QV * NL + i] = 0.;
            }
            else
            {
                const double r1 = sg[G_S1 * NL + i] / sg[G_Z1 * NL + i], r2 = sg[G_S2 * NL + i] / sg[G_Z2 * NL + i];
                sg[G_EINV * NL + i] = 0.25 * (r1 + r2);
                sg[G_QV * NL + i] = (r1 - r2) / (r1 + r2);
            }
        }
    }
    if (k < K)
        buildHs(c, k, identity);
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

// Reduced KKT solve, part 1: condensed right-hand side.  Input: F_BXW/F_BXD, G_BXNU/G_BXNUB/G_BY and b.
// Output: stage field fBeta (16) and segment field gRho (14) for the sweeps; returns the sigma-row rhs.
__device__ inline double kktPrep(const Ctx &c, bool identity, Glob &g, const Rhs &b, int fBeta, int gRho)
{
    const int k = c.lane, K = c.K;
    g.dz3 = -b.n1;
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, SEGREC, unsigned(k));
        for (int i = 0; i < NL; i++)
        {
            const double d1 = dLP(identity, sg[G_S1 * NL + i], sg[G_Z1 * NL + i]);
            const double d2 = dLP(identity, sg[G_S2 * NL + i], sg[G_Z2 * NL + i]);
            sg[G_DINV * NL + i] = 1. / (d1 + d2);
            const double bnb = sg[G_BXNUB * NL + i] - g.dz3;
            const double btn = sg[G_BXNU * NL + i] - sg[G_QV * NL + i] * bnb;
            sg[G_BNB * NL + i] = bnb;
            sg[G_BTN * NL + i] = btn;
            sg[gRho * NL + i] = sg[G_BY * NL + i] + sg[G_EINV * NL + i] * btn;
        }
    }
    if (k < K)
    {
        const SV st = makeSV(c.st, STREC, unsigned(k));
        const unsigned fm = fixedMask(k, K);
        for (int j = 0; j < NV; j++)
            st[fBeta + j] = (fm & (1u << j)) ? 0. : st[F_BXW + j] - st[F_HDW + j] * st[F_BXD] / st[F_HDD];
    }
    return b.s - g.Hsd * b.ds / g.Hdd;
}

// border column products: schur complement of sigma (after the border column has been solved)
__device__ inline void borderSchur(const Ctx &c, Glob &g)
{
    const int k = c.lane, K = c.K;
    double acc = 0.;
    if (k < K - 1)
        for (int i = 0; i < NL; i++)
            acc += -c.S[k * NX + i] * c.sg[size_t(G_BCL * NL + i) * LANES + k];
    acc = wave_sum(acc);
    g.schur = g.hsig - acc;
}

// part 2: after the sweeps left T_mat^-1 [beta;rho] in fVW / gVL: border correction and recovery of the
// eliminated variables.
__device__ inline void kktFinish(const Ctx &c, bool identity, Glob &g, const Rhs &b, double bts, int fVW, int gVL)
{
    const int k = c.lane, K = c.K;
    double cv = 0.;
    if (k < K - 1)
        for (int i = 0; i < NL; i++)
            cv += -c.S[k * NX + i] * c.sg[size_t(gVL * NL + i) * LANES + k];
    cv = wave_sum(cv);
    g.dsig = (bts - cv) / g.schur;
    g.ddsg = (b.ds - g.Hsd * g.dsig) / g.Hdd;
    double sumnb = 0.;
    if (k < K)
    {
        const SV st = makeSV(c.st, STREC, unsigned(k));
        double acc = 0.;
        for (int j = 0; j < NV; j++)
        {
            const double d = st[fVW + j] - st[F_BCW + j] * g.dsig;
            st[F_DW + j] = d;
            acc += st[F_HDW + j] * d;
        }
        st[F_DDL] = (st[F_BXD] - acc) / st[F_HDD];
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, SEGREC, unsigned(k));
        for (int i = 0; i < NL; i++)
        {
            const double dl = sg[gVL * NL + i] - sg[G_BCL * NL + i] * g.dsig;
            sg[G_DLAM * NL + i] = dl;
            const double dnu = sg[G_EINV * NL + i] * (dl + sg[G_BTN * NL + i]);
            const double dnub = sg[G_BNB * NL + i] * sg[G_DINV * NL + i] - sg[G_QV * NL + i] * dnu;
            sg[G_DNU * NL + i] = dnu;
            sg[G_DNUB * NL + i] = dnub;
            sumnb += dnub;
        }
    }
    sumnb = wave_sum(sumnb);
    const double w3sq = identity ? 1. : g.s3 / g.z3;
    g.dn1 = sumnb - w3sq * g.dz3 - b.rhs3;
}

__device__ inline RhsSpec specBorderPlus(int fBeta, int gRho, int fOut, int gOut)
{
    RhsSpec sp;
    sp.n = 2;
    sp.fBeta = fBeta;
    sp.gRho = gRho;
    sp.fOut = fOut;
    sp.gOut = gOut;
    return sp;
}
__device__ inline RhsSpec specSingle(int fBeta, int gRho, int fOut, int gOut)
{
    RhsSpec sp;
    sp.n = 1;
    sp.fBeta = fBeta;
    sp.gRho = gRho;
    sp.fOut = fOut;
    sp.gOut = gOut;
    return sp;
}

__device__ inline void applyPrimalStep(const Ctx &c, Glob &g, double alpha)
{
    const int k = c.lane, K = c.K;
    if (k < K)
    {
        const SV st = makeSV(c.st, STREC, unsigned(k));
        const unsigned fm = fixedMask(k, K);
#pragma unroll
        for (int j = 0; j < NV; j++)
        {
            const double d = st[F_DW + j];
            st[F_W + j] += (fm & (1u << j)) ? 0. : alpha * d;
        }
        st[F_DL] += alpha * st[F_DDL];
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, SEGREC, unsigned(k));
        for (int i = 0; i < NL; i++)
        {
            sg[G_NU * NL + i] += alpha * sg[G_DNU * NL + i];
            sg[G_NUB * NL + i] += alpha * sg[G_DNUB * NL + i];
        }
    }
    g.sig += alpha * g.dsig;
    g.dsg += alpha * g.ddsg;
    g.n1 += alpha * g.dn1;
}

// affine slacks of everything at the current primal point -> stage field fOut, segment fields g1,g2; scalars
__device__ inline void evalAllSaff(const Ctx &c, const Glob &g, int fOut, int g1, int g2, double &os, double &o3, double *oc)
{
    const int k = c.lane, K = c.K;
    double sumnb = 0.;
    if (k < K)
    {
        const SV st = makeSV(c.st, STREC, unsigned(k));
        saff(c.ip, activeMask(k, K), st + F_W, st[F_DL], st + F_WBAR, st + F_UHAT, st + fOut);
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, SEGREC, unsigned(k));
        for (int i = 0; i < NL; i++)
        {
            const double nu = sg[G_NU * NL + i], nub = sg[G_NUB * NL + i];
            sg[g1 * NL + i] = nub - nu;
            sg[g2 * NL + i] = nub + nu;
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
__device__ inline void bring2cone(const Ctx &c, double gamma, int f, int g1, int g2, double &vs, double &v3, double *vc)
{
    const int k = c.lane, K = c.K;
    double alpha = -gamma;
    if (k < K)
    {
        const unsigned act = activeMask(k, K);
        const SV v = makeSV(c.st, STREC, unsigned(k)) + f;
        for (int cix = 0; cix < NCONE; cix++)
            if (act & (1u << cix))
            {
                const SV r = v + coneOff(cix);
                double nrm = 0.;
                for (int i = 1; i < coneDim(cix); i++)
                    nrm += r[i] * r[i];
                const double cres = r[0] - sqrt(nrm);
                if (cres <= 0. && -cres > alpha)
                    alpha = -cres;
            }
        if ((act & 64u) && v[L1] <= 0. && -v[L1] > alpha)
            alpha = -v[L1];
        if ((act & 128u) && v[L2] <= 0. && -v[L2] > alpha)
            alpha = -v[L2];
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, SEGREC, unsigned(k));
        for (int i = 0; i < NL; i++)
        {
            const double a = sg[g1 * NL + i], b = sg[g2 * NL + i];
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
        const unsigned act = activeMask(k, K);
        const SV v = makeSV(c.st, STREC, unsigned(k)) + f;
        for (int cix = 0; cix < NCONE; cix++)
            if (act & (1u << cix))
                v[coneOff(cix)] += alpha;
        if (act & 64u)
            v[L1] += alpha;
        if (act & 128u)
            v[L2] += alpha;
    }
    if (k < K - 1)
    {
        const SV sg = makeSV(c.sg, SEGREC, unsigned(k));
        for (int i = 0; i < NL; i++)
        {
            sg[g1 * NL + i] += alpha;
            sg[g2 * NL + i] += alpha;
        }
    }
    vs += alpha;
    v3 += alpha;
    vc[0] += alpha;
}


// ---- per-cone work on register arrays (compile-time offset / dimension) ----
template <int OFF, int D>
__device__ inline void ldv(const SV &st, int f, double (&v)[D])
{
#pragma unroll
    for (int i = 0; i < D; i++)
        v[i] = st[f + OFF + i];
}
template <int OFF, int D>
__device__ inline void stv(const SV &st, int f, const double (&v)[D])
{
#pragma unroll
    for (int i = 0; i < D; i++)
        st[f + OFF + i] = v[i];
}
// NT scaling of one cone + lambda = W z ; returns 1 if the iterate left the cone
template <int OFF, int D>
__device__ inline int coneScaling(const SV &st, int cix)
{
    double s[D], z[D], w[D], ls[D], eta;
    ldv<OFF, D>(st, F_S, s);
    ldv<OFF, D>(st, F_Z, z);
    if (!cone::nt_scalingS<D>(s, z, eta, w))
        return 1;
    cone::applyWS<D>(eta, w, z, ls);
    st[F_ETA + cix] = eta;
    stv<OFF, D>(st, F_WB, w);
    stv<OFF, D>(st, F_LS, ls);
    return 0;
}
// t = W^-2 rz' + W^-1(lambda \ ds) of one cone
template <int OFF, int D>
__device__ inline void coneT(const SV &st, int cix, int pass, double om, double sigmu)
{
    double w[D], rz[D], b2[D], t[D];
    const double eta = st[F_ETA + cix];
    ldv<OFF, D>(st, F_WB, w);
    ldv<OFF, D>(st, F_RZ, rz);
#pragma unroll
    for (int i = 0; i < D; i++)
        rz[i] *= om;
    cone::applyWinv2S<D>(eta, w, rz, b2);
    if (pass == 0)
    {
        double z[D];
        ldv<OFF, D>(st, F_Z, z);
#pragma unroll
        for (int i = 0; i < D; i++)
            t[i] = b2[i] - z[i];
    }
    else
    {
        double dss[D], dzs[D], ls[D], dsv[D], aa[D];
        ldv<OFF, D>(st, F_DSS, dss);
        ldv<OFF, D>(st, F_DZS, dzs);
        ldv<OFF, D>(st, F_LS, ls);
        cone::conicProductS<D>(dss, dzs, dsv);
#pragma unroll
        for (int i = 0; i < D; i++)
            dsv[i] = -dsv[i];
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
    stv<OFF, D>(st, F_TZ, t);
}
// dz = -W^-2 L dx + t ; ds = -rz' + L dx ; scaled directions ; returns 1/alpha_max of this cone
template <int OFF, int D>
__device__ inline double coneDir(const SV &st, int cix, double om, const double *Ldall)
{
    double w[D], Ld[D], aa[D], t[D], rz[D], dz[D], ds[D], dss[D], dzs[D], ls[D];
    const double eta = st[F_ETA + cix];
    ldv<OFF, D>(st, F_WB, w);
    ldv<OFF, D>(st, F_TZ, t);
    ldv<OFF, D>(st, F_RZ, rz);
    ldv<OFF, D>(st, F_LS, ls);
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
    stv<OFF, D>(st, F_DZ, dz);
    stv<OFF, D>(st, F_DS, ds);
    stv<OFF, D>(st, F_DSS, dss);
    stv<OFF, D>(st, F_DZS, dzs);
    const double a1 = cone::stepInvS<D>(ls, dss), a2 = cone::stepInvS<D>(ls, dzs);
    return a1 > a2 ? a1 : a2;
}
template <int OFF, int D>
__device__ inline void zeroT(const SV &st)
{
#pragma unroll
    for (int i = 0; i < D; i++)
        st[F_TZ + OFF + i] = 0.;
}

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
__global__ void __launch_bounds__(WAVE, IPM_WAVES_PER_SIMD) ipm_kernel(KernelArgs a)
{
#ifdef IPM_PROFILE
    double prof[12] = {0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.};
    const long long t_kernel0 = clock64();
#endif
    const int inst = blockIdx.x;
    if (inst >= a.B)
        return;
    if (a.active && a.active[inst] == 0)
        return;
    __shared__ TileShared sh;
    const int K = a.K, lane = threadIdx.x, k = lane;
    Ctx c;
    c.K = K;
    c.lane = lane;
    double *ws = a.ws + size_t(inst) * workspaceDoubles(K);
    c.st = ws;
    c.sg = ws + size_t(LANES) * STREC;
    c.fac = c.sg + size_t(LANES) * SEGREC;
    c.sv = c.fac + size_t(K) * FACREC;
    c.A = a.A + size_t(inst) * (K - 1) * NX * NX;
    c.B = a.Bm + size_t(inst) * (K - 1) * NX * NU;
    c.C = a.C + size_t(inst) * (K - 1) * NX * NU;
    c.S = a.S + size_t(inst) * (K - 1) * NX;
    c.Z = a.Z + size_t(inst) * (K - 1) * NX;
    c.ip = a.ip + size_t(inst) * IP_N;
    // the out-of-line sweeps take the context by reference; give them their own copy so that `c` never
    // escapes and its pointers stay in SGPRs for the inlined lane=stage phases
    Ctx cesc = c;
    const double *ip = c.ip;
    const double wtrx = a.wtrx[inst];
    const double w_t = ip[IP_WT], w_trt = ip[IP_WTRT], w_vc = ip[IP_WVC];
    const double sigbar = a.sigma[inst];
    const Settings opt = a.opt;
    const unsigned fm = (k < K) ? fixedMask(k, K) : 0u, act = (k < K) ? activeMask(k, K) : 0u;
    const SV st = makeSV(c.st, STREC, unsigned(k < K ? k : 0));
    const SV stN = makeSV(c.st, STREC, unsigned(k < K - 1 ? k + 1 : 0)); // next stage's record
    const SV sg = makeSV(c.sg, SEGREC, unsigned(k < K - 1 ? k : 0));
    const bool vst = k < K, vsg = k < K - 1;

    Glob g;
    g.sig = g.dsg = g.n1 = 0.;
    g.sigbar = sigbar;
    g.ss = g.zs = g.s3 = g.z3 = 1.;
    for (int i = 0; i < 3; i++)
        g.sc3[i] = g.zc3[i] = 0.;
    g.seta = 1.;
    g.sw[0] = 1.;
    g.sw[1] = g.sw[2] = 0.;

    // ---- stage setup: trust-region centre, fixed values, zero start ----
    int Dcount = 0;
    // device memory is not zero-initialised: clear this lane's records (entries of inactive cones are
    // never written afterwards but are swept by the vector updates)
    if (vst)
        for (int i = 0; i < STREC; i++)
            st[i] = 0.;
    if (vsg)
        for (int i = 0; i < SEGREC; i++)
            sg[i] = 0.;
    if (vst)
    {
        const double *Xb = a.X + (size_t(inst) * K + k) * NX, *Ub = a.U + (size_t(inst) * K + k) * NU;
        for (int j = 0; j < 13; j++)
            st[F_WBAR + j] = Xb[j];
        for (int j = 0; j < 3; j++)
            st[F_WBAR + 13 + j] = Ub[j];
        for (int j = 0; j < 3; j++)
            st[F_UHAT + j] = a.uhat[(size_t(inst) * K + k) * 3 + j];
        for (int j = 0; j < NV; j++)
            st[F_W + j] = 0.;
        if (k == 0)
            for (int j = 0; j < 13; j++)
                st[F_W + j] = ip[IP_XINIT + j];
        if (k == K - 1)
            for (int j = 0; j < 13; j++)
                if (fm & (1u << j))
                    st[F_W + j] = ip[IP_XFINAL + j];
        st[F_DL] = 0.;
        for (int b = 0; b < 8; b++)
            if (act & (1u << b))
                Dcount++;
        // identity scalings
        for (int i = 0; i < 6; i++)
            st[F_ETA + i] = 1.;
        for (int i = 0; i < 33; i++)
            st[F_WB + i] = 0.;
        for (int cix = 0; cix < NCONE; cix++)
            st[F_WB + coneOff(cix)] = 1.;
    }
    if (vsg)
    {
        for (int i = 0; i < NL; i++)
        {
            sg[G_NU * NL + i] = 0.;
            sg[G_NUB * NL + i] = 0.;
        }
        Dcount += 2 * NL;
    }
    {
        double dsum = wave_sum(double(Dcount));
        Dcount = int(dsum + 0.5) + 3;
    }
    const int D = Dcount;
    WAVE_SYNC();

    // =============== initialisation (ECOS init, W = I) ===============
    PROF_T(tp0);
    prepareFactor(c, true, g);
    {
        // primal: bx = -L' saff(x0), by = -ry(x0), rhs3 = n1
        Rhs b;
        if (vst)
        {
            double r[NS], gw[NV], gdl;
            saff(ip, act, st + F_W, st[F_DL], st + F_WBAR, st + F_UHAT, r);
            LTmul(ip, fm, r, st + F_UHAT, gw, &gdl);
            for (int j = 0; j < NV; j++)
                st[F_BXW + j] = -gw[j];
            st[F_BXD] = -gdl;
        }
        if (vsg)
        {
            double res[NL];
            dynRes(c, k, st + F_W, stN + F_W, sg + G_NU * NL, g.sig, res);
            for (int i = 0; i < NL; i++)
            {
                sg[G_BXNU * NL + i] = 0.;
                sg[G_BXNUB * NL + i] = 0.;
                sg[G_BY * NL + i] = -res[i];
            }
        }
        b.s = -((g.sig - 0.001) + (g.sig - sigbar));
        b.ds = -(0.5 * (0.5 + 0.5 * g.dsg) - 0.5 * (0.5 - 0.5 * g.dsg));
        b.n1 = 0.;
        b.rhs3 = g.n1;
        const double bts = kktPrep(c, true, g, b, F_BETA, G_RHO);
        WAVE_SYNC();
        const RhsSpec sp = specBorderPlus(F_BETA, G_RHO, F_VW, G_VL);
        factorSweepFused(cesc, sh, sp);
        bwdSweep(cesc, sp);
        borderSchur(c, g);
        kktFinish(c, true, g, b, bts, F_VW, G_VL);
        PROF_T(tp1);
        PROF_ADD(0, tp0, tp1);
        applyPrimalStep(c, g, 1.);
        WAVE_SYNC();
        evalAllSaff(c, g, F_S, G_S1, G_S2, g.ss, g.s3, g.sc3);
        bring2cone(c, opt.gamma, F_S, G_S1, G_S2, g.ss, g.s3, g.sc3);
    }
    {
        // dual: H x' + A'y = -c ; z = -L x'
        Rhs b;
        if (vst)
        {
            for (int j = 0; j < NV; j++)
                st[F_BXW + j] = 0.;
            st[F_BXD] = -wtrx;
        }
        if (vsg)
            for (int i = 0; i < NL; i++)
            {
                sg[G_BXNU * NL + i] = 0.;
                sg[G_BXNUB * NL + i] = 0.;
                sg[G_BY * NL + i] = 0.;
            }
        b.s = -w_t;
        b.ds = -w_trt;
        b.n1 = -w_vc;
        b.rhs3 = 0.;
        const double bts = kktPrep(c, true, g, b, F_BETA, G_RHO);
        WAVE_SYNC();
        const RhsSpec sp = specSingle(F_BETA, G_RHO, F_VW, G_VL);
        fwdSweep(cesc, sp);
        bwdSweep(cesc, sp);
        kktFinish(c, true, g, b, bts, F_VW, G_VL);
        if (vst)
        {
            double t[NS];
            Lmul(ip, act, st + F_DW, st[F_DDL], st + F_UHAT, t);
            for (int i = 0; i < NS; i++)
                st[F_Z + i] = -t[i];
        }
        if (vsg)
            for (int i = 0; i < NL; i++)
            {
                sg[G_LAM * NL + i] = sg[G_DLAM * NL + i];
                const double dnu = sg[G_DNU * NL + i], dnub = sg[G_DNUB * NL + i];
                sg[G_Z1 * NL + i] = -(dnub - dnu);
                sg[G_Z2 * NL + i] = -(dnub + dnu);
            }
        g.zs = -g.dsig;
        g.z3 = g.dz3;
        g.zc3[0] = -0.5 * g.ddsg;
        g.zc3[1] = 0.5 * g.ddsg;
        g.zc3[2] = -g.dsig;
        bring2cone(c, opt.gamma, F_Z, G_Z1, G_Z2, g.zs, g.z3, g.zc3);
    }
    WAVE_SYNC();

    // ---- data norms for the termination test ----
    double resx0, resy0, resz0;
    {
        resx0 = sqrt(K * wtrx * wtrx + w_t * w_t + w_trt * w_trt + w_vc * w_vc);
        resx0 = resx0 > 1. ? resx0 : 1.;
        double nb = 0., nh = 0.;
        double w0[NV], w1[NV];
        if (vst)
        {
            for (int j = 0; j < NV; j++)
                w0[j] = (fm & (1u << j)) ? st[F_W + j] : 0.;
            double r[NS];
            saff(ip, act, w0, 0., st + F_WBAR, st + F_UHAT, r);
            for (int i = 0; i < NS; i++)
                nh += r[i] * r[i];
        }
        if (vsg)
        {
            const unsigned fmn = fixedMask(k + 1, K);
            for (int j = 0; j < NV; j++)
                w1[j] = (fmn & (1u << j)) ? stN[F_W + j] : 0.;
            double zero[NL], res[NL];
            for (int i = 0; i < NL; i++)
                zero[i] = 0.;
            dynRes(c, k, w0, w1, zero, 0., res);
            for (int i = 0; i < NL; i++)
                nb += res[i] * res[i];
        }
        nb = wave_sum(nb);
        nh = wave_sum(nh) + 0.001 * 0.001 + 0.25 + 0.25 + sigbar * sigbar;
        resy0 = sqrt(nb) > 1. ? sqrt(nb) : 1.;
        resz0 = sqrt(nh) > 1. ? sqrt(nh) : 1.;
    }

    int status = -1, iter = 0;
    double pres = 0., dres = 0., gap = 0., pcost = 0.;
    double rzs = 0., rz3 = 0., rzc[3] = {0., 0., 0.}, rxs = 0., rxds = 0., rxn1 = 0.;
    for (iter = 0;; iter++)
    {
        // ================= residuals =================
        PROF_T(tr0);
        double sas, sa3, sac[3];
        evalAllSaff(c, g, F_RZ, G_RZ1, G_RZ2, sas, sa3, sac);
        double p_gap = 0., p_rx = 0., p_ry = 0., p_rz = 0., p_xx = 0., p_yy = 0., p_zz = 0., p_ss = 0., p_rxs = 0., p_dl = 0.;
        if (vst)
        {
            for (int i = 0; i < NS; i++)
            {
                const double sv = st[F_S + i], zv = st[F_Z + i];
                const double r = sv - st[F_RZ + i];
                st[F_RZ + i] = r;
                p_gap += sv * zv;
                p_rz += r * r;
                p_zz += zv * zv;
                p_ss += sv * sv;
            }
            double gw[NV], gdl;
            LTmul(ip, fm, st + F_Z, st + F_UHAT, gw, &gdl);
            const double rxd = wtrx - gdl;
            st[F_RXD] = rxd;
            double r[NV];
            for (int j = 0; j < NV; j++)
                r[j] = -gw[j];
            if (vsg)
                for (int i = 0; i < NL; i++)
                {
                    const double l = sg[G_LAM * NL + i];
                    for (int j = 0; j < NV; j++)
                        r[j] += Ment(c, k, fm, i, j) * l;
                }
            if (k > 0)
                for (int i = 0; i < NL; i++)
                {
                    const double l = makeSV(c.sg, SEGREC, unsigned(k - 1))[G_LAM * NL + i];
                    for (int j = 0; j < NV; j++)
                        r[j] += Nent(c, k - 1, fm, i, j) * l;
                }
            p_rx += rxd * rxd;
            p_xx += st[F_DL] * st[F_DL];
            p_dl += st[F_DL];
            for (int j = 0; j < NV; j++)
            {
                st[F_RXW + j] = r[j];
                if (!(fm & (1u << j)))
                {
                    p_rx += r[j] * r[j];
                    p_xx += st[F_W + j] * st[F_W + j];
                }
            }
        }
        if (vsg)
        {
            double res[NL];
            dynRes(c, k, st + F_W, stN + F_W, sg + G_NU * NL, g.sig, res);
            for (int i = 0; i < NL; i++)
            {
                const double s1 = sg[G_S1 * NL + i], z1 = sg[G_Z1 * NL + i], s2 = sg[G_S2 * NL + i], z2 = sg[G_Z2 * NL + i];
                const double r1 = s1 - sg[G_RZ1 * NL + i], r2 = s2 - sg[G_RZ2 * NL + i];
                sg[G_RZ1 * NL + i] = r1;
                sg[G_RZ2 * NL + i] = r2;
                sg[G_RY * NL + i] = res[i];
                const double l = sg[G_LAM * NL + i];
                const double rnu = -l + z1 - z2, rnub = -z1 - z2 + g.z3;
                sg[G_RXNU * NL + i] = rnu;
                sg[G_RXNUB * NL + i] = rnub;
                p_gap += s1 * z1 + s2 * z2;
                p_rz += r1 * r1 + r2 * r2;
                p_zz += z1 * z1 + z2 * z2;
                p_ss += s1 * s1 + s2 * s2;
                p_ry += res[i] * res[i];
                p_yy += l * l;
                p_rx += rnu * rnu + rnub * rnub;
                const double nu = sg[G_NU * NL + i], nub = sg[G_NUB * NL + i];
                p_xx += nu * nu + nub * nub;
                p_rxs += c.S[k * NX + i] * l;
            }
        }
        p_gap = wave_sum(p_gap);
        p_rx = wave_sum(p_rx);
        p_ry = wave_sum(p_ry);
        p_rz = wave_sum(p_rz);
        p_xx = wave_sum(p_xx);
        p_yy = wave_sum(p_yy);
        p_zz = wave_sum(p_zz);
        p_ss = wave_sum(p_ss);
        p_rxs = wave_sum(p_rxs);
        p_dl = wave_sum(p_dl);
        rzs = g.ss - sas;
        rz3 = g.s3 - sa3;
        for (int i = 0; i < 3; i++)
            rzc[i] = g.sc3[i] - sac[i];
        rxs = w_t - g.zs - g.zc3[2] - p_rxs;
        rxds = w_trt - 0.5 * g.zc3[0] + 0.5 * g.zc3[1];
        rxn1 = w_vc - g.z3;
        gap = p_gap + g.ss * g.zs + g.s3 * g.z3;
        double nrz = p_rz + rzs * rzs + rz3 * rz3;
        double nzz = p_zz + g.zs * g.zs + g.z3 * g.z3;
        double nss = p_ss + g.ss * g.ss + g.s3 * g.s3;
        for (int i = 0; i < 3; i++)
        {
            gap += g.sc3[i] * g.zc3[i];
            nrz += rzc[i] * rzc[i];
            nzz += g.zc3[i] * g.zc3[i];
            nss += g.sc3[i] * g.sc3[i];
        }
        const double nrx = p_rx + rxs * rxs + rxds * rxds + rxn1 * rxn1;
        const double nxx = p_xx + g.sig * g.sig + g.dsg * g.dsg + g.n1 * g.n1;
        const double mu = gap / D;
        pcost = w_t * g.sig + w_trt * g.dsg + w_vc * g.n1 + wtrx * p_dl;
        {
            const double nx_ = sqrt(nxx), ny_ = sqrt(p_yy), nz_ = sqrt(nzz), ns_ = sqrt(nss);
            const double d1 = resy0 + nx_ > 1. ? resy0 + nx_ : 1.;
            const double d2 = resz0 + nx_ + ns_ > 1. ? resz0 + nx_ + ns_ : 1.;
            const double pa = sqrt(p_ry) / d1, pb = sqrt(nrz) / d2;
            pres = pa > pb ? pa : pb;
            const double d3 = resx0 + ny_ + nz_ > 1. ? resx0 + ny_ + nz_ : 1.;
            dres = sqrt(nrx) / d3;
        }
        const double apc = fabs(pcost) > 1e-300 ? fabs(pcost) : 1e-300;
        const double relgap = gap / apc;
        if (!(pres == pres) || !(dres == dres) || !(gap == gap) || fabs(pres) > 1e300 || fabs(dres) > 1e300 || fabs(gap) > 1e300)
        {
            status = -2;
            break;
        }
        if (pres < opt.feastol && dres < opt.feastol && (gap < opt.abstol || relgap < opt.reltol))
        {
            status = 0;
            break;
        }
        if (iter >= opt.maxit)
        {
            status = -1;
            break;
        }

        // ================= scalings =================
        PROF_T(tr1);
        PROF_ADD(1, tr0, tr1);
        int bad = 0;
        if (vst)
        {
            bad |= coneScaling<C1, 17>(st, 0);
            if (act & 2u)
                bad |= coneScaling<C2, 3>(st, 1);
            if (act & 4u)
                bad |= coneScaling<C3, 3>(st, 2);
            if (act & 8u)
                bad |= coneScaling<C4, 3>(st, 3);
            bad |= coneScaling<C5, 4>(st, 4);
            bad |= coneScaling<C6, 3>(st, 5);
        }
        if (!cone::nt_scaling(g.sc3, g.zc3, 3, g.seta, g.sw))
            bad = 1;
        bad = wave_or(bad);
        if (bad)
        {
            status = -2;
            break;
        }
        cone::applyW(g.seta, g.sw, 3, g.zc3, g.lamC);
        WAVE_SYNC();
        PROF_T(tr2);
        PROF_ADD(2, tr1, tr2);
        prepareFactor(c, false, g);
        PROF_T(tr3);
        PROF_ADD(8, tr2, tr3);

        double sigma_c = 0., alpha = 1.;
        double tzs = 0., tzc[3] = {0., 0., 0.};
        for (int pass = 0; pass < 2 && !bad; pass++)
        {
            const double om = 1. - sigma_c;
            PROF_T(tq0);
            // ---------- t = W^-2 rz' + W^-1(lambda \ ds) ; bx = -rx' + L't ----------
            if (vst)
            {
                {
                    const double sigmu = sigma_c * mu;
                    coneT<C1, 17>(st, 0, pass, om, sigmu);
                    if (act & 2u)
                        coneT<C2, 3>(st, 1, pass, om, sigmu);
                    else
                        zeroT<C2, 3>(st);
                    if (act & 4u)
                        coneT<C3, 3>(st, 2, pass, om, sigmu);
                    else
                        zeroT<C3, 3>(st);
                    if (act & 8u)
                        coneT<C4, 3>(st, 3, pass, om, sigmu);
                    else
                        zeroT<C4, 3>(st);
                    coneT<C5, 4>(st, 4, pass, om, sigmu);
                    coneT<C6, 3>(st, 5, pass, om, sigmu);
                }
                for (int which = 0; which < 2; which++)
                {
                    const int o = which ? L2 : L1;
                    if (!(act & (1u << (6 + which))))
                    {
                        st[F_TZ + o] = 0.;
                        continue;
                    }
                    const double sv = st[F_S + o], zv = st[F_Z + o];
                    const double corr = pass ? (sigma_c * mu - st[F_DS + o] * st[F_DZ + o]) / sv : 0.;
                    st[F_TZ + o] = (zv / sv) * om * st[F_RZ + o] - zv + corr;
                }
                double gw[NV], gdl;
                LTmul(ip, fm, st + F_TZ, st + F_UHAT, gw, &gdl);
                for (int j = 0; j < NV; j++)
                    st[F_BXW + j] = -om * st[F_RXW + j] + gw[j];
                st[F_BXD] = -om * st[F_RXD] + gdl;
            }
            if (vsg)
                for (int i = 0; i < NL; i++)
                {
                    const double s1 = sg[G_S1 * NL + i], z1 = sg[G_Z1 * NL + i], s2 = sg[G_S2 * NL + i], z2 = sg[G_Z2 * NL + i];
                    const double c1 = pass ? (sigma_c * mu - sg[G_DS1 * NL + i] * sg[G_DZ1 * NL + i]) / s1 : 0.;
                    const double c2 = pass ? (sigma_c * mu - sg[G_DS2 * NL + i] * sg[G_DZ2 * NL + i]) / s2 : 0.;
                    const double t1 = (z1 / s1) * om * sg[G_RZ1 * NL + i] - z1 + c1;
                    const double t2 = (z2 / s2) * om * sg[G_RZ2 * NL + i] - z2 + c2;
                    sg[G_TZ1 * NL + i] = t1;
                    sg[G_TZ2 * NL + i] = t2;
                    sg[G_BXNU * NL + i] = -om * sg[G_RXNU * NL + i] + (-t1 + t2);
                    sg[G_BXNUB * NL + i] = -om * sg[G_RXNUB * NL + i] + (t1 + t2);
                    sg[G_BY * NL + i] = -om * sg[G_RY * NL + i];
                }
            tzs = (g.zs / g.ss) * om * rzs - g.zs + (pass ? (sigma_c * mu - g.dss * g.dzs) / g.ss : 0.);
            {
                double aa[3], b2[3];
                for (int i = 0; i < 3; i++)
                    aa[i] = om * rzc[i];
                cone::applyWinv2(g.seta, g.sw, 3, aa, b2);
                if (pass == 0)
                    for (int i = 0; i < 3; i++)
                        tzc[i] = b2[i] - g.zc3[i];
                else
                {
                    double dsv[3];
                    cone::conicProduct(3, g.dsC, g.dzC, dsv);
                    for (int i = 0; i < 3; i++)
                        dsv[i] = -dsv[i];
                    dsv[0] += sigma_c * mu;
                    cone::conicDivision(3, g.lamC, dsv, dsv);
                    for (int i = 0; i < 3; i++)
                        dsv[i] -= g.lamC[i];
                    cone::applyWinv(g.seta, g.sw, 3, dsv, aa);
                    for (int i = 0; i < 3; i++)
                        tzc[i] = b2[i] + aa[i];
                }
            }
            const double ds3v = -g.s3 * g.z3 + (pass ? (sigma_c * mu - g.ds3 * g.dz3) : 0.);
            Rhs b;
            b.s = -om * rxs + tzs + tzc[2];
            b.ds = -om * rxds + 0.5 * tzc[0] - 0.5 * tzc[1];
            b.n1 = -om * rxn1;
            b.rhs3 = -om * rz3 - ds3v / g.z3;
            const double bts = kktPrep(c, false, g, b, F_BETA, G_RHO);
            WAVE_SYNC();
            PROF_T(tq1);
            PROF_ADD(4, tq0, tq1);
            if (pass == 0)
            {
                // one factorisation per iteration, fused with the forward substitution of the sigma border
                // column and of the affine right-hand side
                const RhsSpec sp = specBorderPlus(F_BETA, G_RHO, F_VW, G_VL);
                factorSweepFused(cesc, sh, sp);
                PROF_T(tf1);
                PROF_ADD(3, tq1, tf1);
                bwdSweep(cesc, sp);
                borderSchur(c, g);
                PROF_T(tf2);
                PROF_ADD(9, tf1, tf2);
            }
            else
            {
                const RhsSpec sp = specSingle(F_BETA, G_RHO, F_VW, G_VL);
                fwdSweep(cesc, sp);
                PROF_T(tf1);
                PROF_ADD(10, tq1, tf1);
                bwdSweep(cesc, sp);
                PROF_T(tf2);
                PROF_ADD(9, tf1, tf2);
            }
            if (!(g.schur > 0.))
                bad = 1;
            kktFinish(c, false, g, b, bts, F_VW, G_VL);
            PROF_T(tq2);
            PROF_ADD(5, tq1, tq2);
            // ---------- dz = -W^-2 L dx + t ; ds = -rz' + L dx ; step length ----------
            double ainv = 0.;
            if (vst)
            {
                double Ld[NS];
                Lmul(ip, act, st + F_DW, st[F_DDL], st + F_UHAT, Ld);
                {
                    double a0 = coneDir<C1, 17>(st, 0, om, Ld);
                    ainv = a0 > ainv ? a0 : ainv;
                    if (act & 2u)
                    {
                        a0 = coneDir<C2, 3>(st, 1, om, Ld);
                        ainv = a0 > ainv ? a0 : ainv;
                    }
                    if (act & 4u)
                    {
                        a0 = coneDir<C3, 3>(st, 2, om, Ld);
                        ainv = a0 > ainv ? a0 : ainv;
                    }
                    if (act & 8u)
                    {
                        a0 = coneDir<C4, 3>(st, 3, om, Ld);
                        ainv = a0 > ainv ? a0 : ainv;
                    }
                    a0 = coneDir<C5, 4>(st, 4, om, Ld);
                    ainv = a0 > ainv ? a0 : ainv;
                    a0 = coneDir<C6, 3>(st, 5, om, Ld);
                    ainv = a0 > ainv ? a0 : ainv;
                }
                for (int which = 0; which < 2; which++)
                {
                    const int o = which ? L2 : L1;
                    if (!(act & (1u << (6 + which))))
                        continue;
                    const double sv = st[F_S + o], zv = st[F_Z + o];
                    const double dzv = -(zv / sv) * Ld[o] + st[F_TZ + o];
                    const double dsv = -om * st[F_RZ + o] + Ld[o];
                    st[F_DZ + o] = dzv;
                    st[F_DS + o] = dsv;
                    const double a1 = -dsv / sv, a2 = -dzv / zv;
                    ainv = a1 > ainv ? a1 : ainv;
                    ainv = a2 > ainv ? a2 : ainv;
                }
            }
            double sumdnb = 0.;
            if (vsg)
                for (int i = 0; i < NL; i++)
                {
                    const double s1 = sg[G_S1 * NL + i], z1 = sg[G_Z1 * NL + i], s2 = sg[G_S2 * NL + i], z2 = sg[G_Z2 * NL + i];
                    const double dnu = sg[G_DNU * NL + i], dnub = sg[G_DNUB * NL + i];
                    const double L1v = dnub - dnu, L2v = dnub + dnu;
                    const double dz1 = -(z1 / s1) * L1v + sg[G_TZ1 * NL + i], ds1 = -om * sg[G_RZ1 * NL + i] + L1v;
                    const double dz2 = -(z2 / s2) * L2v + sg[G_TZ2 * NL + i], ds2 = -om * sg[G_RZ2 * NL + i] + L2v;
                    sg[G_DZ1 * NL + i] = dz1;
                    sg[G_DS1 * NL + i] = ds1;
                    sg[G_DZ2 * NL + i] = dz2;
                    sg[G_DS2 * NL + i] = ds2;
                    double m1 = -ds1 / s1, m2 = -dz1 / z1, m3 = -ds2 / s2, m4 = -dz2 / z2;
                    m1 = m1 > m2 ? m1 : m2;
                    m3 = m3 > m4 ? m3 : m4;
                    m1 = m1 > m3 ? m1 : m3;
                    ainv = m1 > ainv ? m1 : ainv;
                    sumdnb += dnub;
                }
            sumdnb = wave_sum(sumdnb);
            g.dzs = -(g.zs / g.ss) * g.dsig + tzs;
            g.dss = -om * rzs + g.dsig;
            {
                const double m1 = -g.dss / g.ss, m2 = -g.dzs / g.zs;
                ainv = m1 > ainv ? m1 : ainv;
                ainv = m2 > ainv ? m2 : ainv;
            }
            g.ds3 = -om * rz3 + (g.dn1 - sumdnb);
            {
                const double m1 = -g.ds3 / g.s3, m2 = -g.dz3 / g.z3;
                ainv = m1 > ainv ? m1 : ainv;
                ainv = m2 > ainv ? m2 : ainv;
            }
            {
                const double Ld[3] = {0.5 * g.ddsg, -0.5 * g.ddsg, g.dsig};
                double aa[3];
                cone::applyWinv2(g.seta, g.sw, 3, Ld, aa);
                for (int i = 0; i < 3; i++)
                {
                    g.dzc3[i] = -aa[i] + tzc[i];
                    g.dsc3[i] = -om * rzc[i] + Ld[i];
                }
                cone::applyWinv(g.seta, g.sw, 3, g.dsc3, g.dsC);
                cone::applyW(g.seta, g.sw, 3, g.dzc3, g.dzC);
                const double a1 = cone::stepInv(3, g.lamC, g.dsC), a2 = cone::stepInv(3, g.lamC, g.dzC);
                ainv = a1 > ainv ? a1 : ainv;
                ainv = a2 > ainv ? a2 : ainv;
            }
            ainv = wave_max(ainv);
            if (pass == 0)
            {
                double alpha_a = ainv > 0. ? 1. / ainv : 1.;
                alpha_a = alpha_a < 1. ? alpha_a : 1.;
                sigma_c = (1. - alpha_a) * (1. - alpha_a) * (1. - alpha_a);
                sigma_c = sigma_c < 1e-4 ? 1e-4 : sigma_c;
                sigma_c = sigma_c > 1. ? 1. : sigma_c;
            }
            else
            {
                alpha = ainv > 0. ? opt.gamma / ainv : 1.;
                alpha = alpha < 1. ? alpha : 1.;
                alpha = alpha < 0.999 ? alpha : 0.999;
                alpha = alpha > 1e-8 ? alpha : 1e-8;
            }
            WAVE_SYNC();
            PROF_T(tq3);
            PROF_ADD(6, tq2, tq3);
        }
        PROF_T(tu0);
        if (bad)
        {
            status = -2;
            break;
        }
        // ================= update =================
        applyPrimalStep(c, g, alpha);
        if (vst)
            for (int i = 0; i < NS; i++)
            {
                st[F_S + i] += alpha * st[F_DS + i];
                st[F_Z + i] += alpha * st[F_DZ + i];
            }
        if (vsg)
            for (int i = 0; i < NL; i++)
            {
                sg[G_LAM * NL + i] += alpha * sg[G_DLAM * NL + i];
                sg[G_S1 * NL + i] += alpha * sg[G_DS1 * NL + i];
                sg[G_Z1 * NL + i] += alpha * sg[G_DZ1 * NL + i];
                sg[G_S2 * NL + i] += alpha * sg[G_DS2 * NL + i];
                sg[G_Z2 * NL + i] += alpha * sg[G_DZ2 * NL + i];
            }
        g.ss += alpha * g.dss;
        g.zs += alpha * g.dzs;
        g.s3 += alpha * g.ds3;
        g.z3 += alpha * g.dz3;
        for (int i = 0; i < 3; i++)
        {
            g.sc3[i] += alpha * g.dsc3[i];
            g.zc3[i] += alpha * g.dzc3[i];
        }
        WAVE_SYNC();
        PROF_T(tu1);
        PROF_ADD(7, tu0, tu1);
    }

    // =============== outputs: readSolution + SC bookkeeping ===============
    double sum_delta = 0.;
    if (vst)
        sum_delta = st[F_DL];
    sum_delta = wave_sum(sum_delta);
#ifdef IPM_PROFILE
    if (a.dbg && lane == 0)
    {
        double *d = a.dbg + size_t(inst) * 32;
        prof[11] = double(clock64() - t_kernel0);
        for (int i = 0; i < 12; i++)
            d[8 + i] = prof[i];
    }
#endif
    if (a.dbg && lane == 0)
    {
        double *d = a.dbg + size_t(inst) * 32;
        d[0] = pcost;
        d[1] = gap;
        d[2] = pres;
        d[3] = dres;
        d[4] = iter;
        d[5] = status;
        d[6] = g.n1;
        d[7] = sum_delta;
    }
    if (status == 0)
    {
        if (vst)
        {
            double *Xo = a.X + (size_t(inst) * K + k) * NX, *Uo = a.U + (size_t(inst) * K + k) * NU;
            for (int j = 0; j < 13; j++)
                Xo[j] = st[F_W + j];
            Xo[13] = 0.;
            for (int j = 0; j < 3; j++)
                Uo[j] = st[F_W + 13 + j];
            Uo[3] = 0.;
        }
        if (lane == 0)
            a.sigma[inst] = g.sig;
    }
    if (lane == 0)
    {
        a.ipm_iters[inst] += iter;
        a.norm1_nu[inst] = g.n1;
        a.sum_delta[inst] = sum_delta;
        a.delta_sigma[inst] = g.dsg;
        if (a.do_sc_update)
        {
            a.sc_iters[inst] += 1;
            if (status != 0)
            {
                a.status[inst] = status; // solver failure: reference would std::terminate (SCAlgorithm.cpp:94-98)
                a.active[inst] = 0;
            }
            else
            {
                if (g.n1 < a.nu_tol)
                    a.wtrx[inst] = wtrx * 2.;
                const int conv = (sum_delta < a.delta_tol && g.n1 < a.nu_tol) ? 1 : 0;
                if (conv)
                {
                    a.converged[inst] = 1;
                    a.active[inst] = 0;
                }
                else if (a.sc_iters[inst] >= a.max_sc_iterations)
                    a.active[inst] = 0;
            }
        }
        else
            a.status[inst] = status;
    }
}

} // namespace ipm
} // namespace scpp
