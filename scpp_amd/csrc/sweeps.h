// Stage sweeps of the block-tridiagonal KKT system on the tile engine (see tile_engine.h for the algebra).
//   factorSweepFused : factorisation + forward substitution of up to 2 right-hand-side columns
//   fwdSweep         : forward substitution only (re-uses the stored tiles)
//   bwdSweep         : backward substitution, writes dw / dlam columns to the exchange records
// Right-hand sides and solutions live in the stage-major exchange records (Lay<P>::X_*).  Column 0 of a 2-column
// sweep is the sigma BORDER column (beta = 0, rho = -S_k, result to X_BCW / X_BCL); the other column reads
// beta from X_BETA (16 entries) and rho from X_RHO (NL entries) and leaves its solution in X_VW / X_VL.
// All sweeps are software-pipelined: the global loads of stage k+1 (k-1) are issued before the dependent
// MFMA / elimination chain of stage k so that HBM/L2 latency overlaps the chain.
#pragma once
#include "tile_engine.h"

namespace scpp
{
namespace ipm
{

#ifndef SWEEPS_INLINE
#define SWEEP_FN static __device__ __attribute__((noinline, disable_tail_calls))
#else
#define SWEEP_FN __device__ inline __attribute__((always_inline))
#endif

// the instance context lives in LDS (one copy per wavefront); out-of-line phases / sweeps re-materialise it in SGPRs
#ifdef SCPP_HIP_EMU
#define LDSP
#else
#define LDSP __attribute__((address_space(3)))
#endif
__device__ inline Ctx uniformCtx(const LDSP Ctx *cin)
{
    Ctx c;
    c.K = uniformInt(cin->K);
    c.lane = threadIdx.x;
    c.pitch = uniformInt(cin->pitch);
    c.st = uniformPtr(cin->st);
    c.sg = uniformPtr(cin->sg);
    c.dy = uniformPtr(cin->dy);
    c.fac = uniformPtr(cin->fac);
    c.sv = uniformPtr(cin->sv);
    c.sx = uniformPtr(cin->sx);
    c.gsave = uniformPtr(cin->gsave);
    c.A = uniformPtr(cin->A);
    c.B = uniformPtr(cin->B);
    c.C = uniformPtr(cin->C);
    c.S = uniformPtr(cin->S);
    c.Z = uniformPtr(cin->Z);
    c.ip = uniformPtr(cin->ip);
    return c;
}
// the exchange records through a buffer resource: stage offset in an SGPR, the lane's entry in the VGPR / immediate offset,
// so that a sweep carries no 64-bit per-lane addresses for them
struct XS
{
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ double ld(int stage_bytes, int entry) const
    {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, entry * 8, stage_bytes, 0));
    }
    __device__ void st(int stage_bytes, int entry, double x) const
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, x), rsrc, entry * 8, stage_bytes, 0);
    }
};
template <class P>
__device__ inline XS makeXS(const Ctx &c)
{
    return XS{__builtin_amdgcn_make_buffer_rsrc(c.sx, 0, c.K * Lay<P>::XREC * 8, 0x00020000)};
}
template <class P>
__device__ inline int xsStage(int k)
{
    return k * (Lay<P>::XREC * 8);
}
struct RhsSpec
{
    int n; // 1: single column (X_BETA / X_RHO -> X_VW / X_VL) ; 2: [border | column] (border: -S_k -> X_BCW / X_BCL)
};

__device__ inline RhsSpec uniformSpec(const RhsSpec &s)
{
    RhsSpec o;
    o.n = uniformInt(s.n);
    return o;
}

// column index of the regular (non-border) column, -1 if this lane's column is unused
__device__ inline int colKind(const RhsSpec &sp, int i) // 0 none, 1 border, 2 regular
{
    if (sp.n == 2)
        return i == 0 ? 1 : (i == 1 ? 2 : 0);
    return i == 0 ? 2 : 0;
}

template <class P>
__device__ inline Tile loadRhsW(const XS &xs, const RhsSpec &sp, int k, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Tile t = tileZero();
    if (colKind(sp, i) == 2)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
            t.v[r] = xs.ld(xsStage<P>(k), Lay<P>::X_BETA + g + 4 * r);
    }
    return t;
}
template <class P>
__device__ inline Tile loadRhsL(const Ctx &c, const XS &xs, const RhsSpec &sp, int k, int lane)
{
    constexpr int NL = Lay<P>::NL, NX = P::NX;
    const int g = lane >> 4, i = lane & 15;
    Tile t = tileZero();
    const int kind = colKind(sp, i);
    if (kind)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int row = g + 4 * r;
            if (row < NL)
                t.v[r] = kind == 1 ? c.S[k * NX + row] : xs.ld(xsStage<P>(k), Lay<P>::X_RHO + row);
        }
    }
    return t;
}
// border column stores -S: sign applied at use so that the load itself carries no arithmetic
__device__ inline Tile rhsLSign(const RhsSpec &sp, int lane, const Tile &t)
{
    const int i = lane & 15;
    if (colKind(sp, i) == 1)
    {
        Tile o;
#pragma unroll
        for (int r = 0; r < 4; r++)
            o.v[r] = -t.v[r];
        return o;
    }
    return t;
}
__device__ inline void saveCols(double *p, int n, int lane, const Tile &t)
{
    const int g = lane >> 4, i = lane & 15;
    if (i < n)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
            p[i * 16 + g + 4 * r] = t.v[r];
    }
}
__device__ inline Tile loadCols(const double *p, int n, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Tile t = tileZero();
    if (i < n)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
            t.v[r] = p[i * 16 + g + 4 * r];
    }
    return t;
}

__device__ inline Tile tileSub(const Tile &a, const Tile &b)
{
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
        t.v[r] = a.v[r] - b.v[r];
    return t;
}
__device__ inline Tile tileAdd(const Tile &a, const Tile &b)
{
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
        t.v[r] = a.v[r] + b.v[r];
    return t;
}

// ---- raw (arithmetic-free) stage inputs of the factorisation ----
struct HRaw
{
    double e2, cc, wcol, wrow[4], hs[4];
};
// position of this lane's four tile entries (g + 4r, i) in the small-block record (-1: outside the pattern); depends on the
// lane only, so a sweep looks it up once instead of once per stage
struct HsLane
{
    int idx[4];
};
template <class P>
__device__ inline HsLane hsLane(int lane)
{
    const int g = lane >> 4, i = lane & 15;
    HsLane h;
#pragma unroll
    for (int r = 0; r < 4; r++)
        h.idx[r] = hsIndex<P>(g + 4 * r, i);
    return h;
}
template <class P>
__device__ inline HRaw loadHRaw(const XS &xs, const HsLane &hl, int k, int lane)
{
    using L = Lay<P>;
    const int g = lane >> 4, i = lane & 15;
    const int sk = xsStage<P>(k);
    HRaw h;
    h.e2 = xs.ld(sk, L::X_HC);
    h.cc = xs.ld(sk, L::X_HC + 1);
    h.wcol = xs.ld(sk, L::X_WBT + i);
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        h.wrow[r] = xs.ld(sk, L::X_WBT + row);
        h.hs[r] = xs.ld(sk, L::X_HS + (hl.idx[r] >= 0 ? hl.idx[r] : 0));
    }
    return h;
}
// H_k (16x16, delta_k eliminated, fixed variables -> identity rows) as a D-layout tile
template <class P>
__device__ inline Tile buildHTile(const HRaw &h, const HsLane &hl, int k, int K, int lane, bool scvx)
{
    const int g = lane >> 4, i = lane & 15;
    const unsigned fm = Lay<P>::fixedMask(k, K);
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        double v = h.e2 * ((row == i ? 1. : 0.) - h.cc * h.wrow[r] * h.wcol);
        if (scvx && (row < P::NXV || i < P::NXV)) // SCvx: the trust cone has no state rows
            v = 0.;
        if (hl.idx[r] >= 0)
            v += h.hs[r];
        if ((fm & (1u << row)) || (fm & (1u << i)))
            v = (row == i) ? 1. : 0.;
        t.v[r] = v;
    }
    return t;
}
// raw entries of [A|B] / C arranged for the M' and N tiles (mask and sign applied at use)
// state / input column a stage variable reads from A / B (lane-dependent index: small constant table)
template <class P>
__device__ inline int varColumn(int j)
{
    using L = Lay<P>;
    if constexpr (L::IDENTITY_MAPS)
        return j < P::NXV ? j : (j < L::NVU ? j - P::NXV : 0); // no table look-up on the sweeps' critical path
    else
        return j < P::NXV ? P::XMAP[j] : (j < L::NVU ? P::UMAP[j - P::NXV] : 0);
}
template <class P>
__device__ inline Tile loadMtRaw(const Ctx &c, int k, int lane) // entry (var j = g+4r, dyn row i)
{
    using L = Lay<P>;
    constexpr int NX = P::NX, NU = P::NU;
    const int g = lane >> 4, i = lane & 15;
    Tile t = tileZero();
    if (i < L::NL)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int j = g + 4 * r;
            if (j < L::NVU)
                t.v[r] = j < P::NXV ? c.A[size_t(k) * NX * NX + i * NX + varColumn<P>(j)] : c.B[size_t(k) * NX * NU + i * NU + varColumn<P>(j)];
        }
    }
    return t;
}
__device__ inline Tile finishMt(const Tile &raw, unsigned fm, int lane)
{
    const int g = lane >> 4;
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
        t.v[r] = (fm & (1u << (g + 4 * r))) ? 0. : -raw.v[r];
    return t;
}
template <class P>
__device__ inline Tile loadNRaw(const Ctx &c, int k, int lane) // entry (dyn row g+4r, var i): only the C part is loaded
{
    using L = Lay<P>;
    constexpr int NX = P::NX, NU = P::NU;
    const int g = lane >> 4, i = lane & 15;
    Tile t = tileZero();
    if (i >= P::NXV && i < L::NVU)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int row = g + 4 * r;
            if (row < L::NL)
                t.v[r] = c.C[size_t(k) * NX * NU + row * NU + varColumn<P>(i)];
        }
    }
    return t;
}
template <class P>
__device__ inline Tile finishN(const Tile &raw, unsigned fmn, int lane)
{
    using L = Lay<P>;
    const int g = lane >> 4, i = lane & 15;
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        double v = 0.;
        if (row < L::NL && !(fmn & (1u << i)))
            v = i < P::NXV ? (row == varColumn<P>(i) ? 1. : 0.) : -raw.v[r];
        t.v[r] = v;
    }
    return t;
}

// N' tile: entry (var a = g+4r, dyn row b = i) = N[b][a]; only the C part is loaded
template <class P>
__device__ inline Tile loadNtRaw(const Ctx &c, int k, int lane)
{
    using L = Lay<P>;
    constexpr int NX = P::NX, NU = P::NU;
    const int g = lane >> 4, i = lane & 15;
    Tile t = tileZero();
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        if (4 * r + 3 < P::NXV || 4 * r >= L::NVU) // this register holds no input variable (RocketQuat: inputs 13..15 live in register 3)
            continue;
        const int a = g + 4 * r;
        const bool in = i < L::NL && a >= P::NXV && a < L::NVU;
        const double v = c.C[size_t(k) * NX * NU + (i < L::NL ? i : 0) * NU + (in ? varColumn<P>(a) : 0)];
        t.v[r] = in ? v : 0.;
    }
    return t;
}
template <class P>
__device__ inline Tile finishNt(const Tile &raw, unsigned fmn, int lane)
{
    using L = Lay<P>;
    const int g = lane >> 4, i = lane & 15;
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int a = g + 4 * r;
        double v = 0.;
        if (i < L::NL && !(fmn & (1u << a)))
            v = a < P::NXV ? (varColumn<P>(a) == i ? 1. : 0.) : -raw.v[r];
        t.v[r] = v;
    }
    return t;
}

// Stage inputs of the factor sweep.  Only the Hessian inputs (needed at the very start of a stage) are prefetched one
// stage ahead; the coupling tiles and right-hand sides are requested at the START of their own stage and consumed after
// the first inverse-factor elimination, whose dependent chain (thousands of cycles) hides their latency -- which keeps
// 34 VGPRs of prefetch buffer out of the elimination's live set.
struct FactorRest
{
    Tile mt, n, rl, rwn;
    double einv;
};
template <class P>
__device__ inline FactorRest loadFactorRest(const Ctx &c, const XS &xs, const RhsSpec &sp, int k, int lane)
{
    constexpr int NL = Lay<P>::NL;
    const int i = lane & 15;
    FactorRest f;
    if (k < c.K - 1)
    {
        f.mt = loadMtRaw<P>(c, k, lane);
        f.n = loadNRaw<P>(c, k, lane);
        f.rl = loadRhsL<P>(c, xs, sp, k, lane);
        f.rwn = loadRhsW<P>(xs, sp, k + 1, lane);
        f.einv = xs.ld(xsStage<P>(k), Lay<P>::X_EINV + (i < NL ? i : 0));
    }
    else
    {
        f.mt = f.n = f.rl = f.rwn = tileZero();
        f.einv = 1.;
    }
    return f;
}

template <class P>
SWEEP_FN void factorSweepFused(const LDSP Ctx *cin, TileShared &sh, const RhsSpec &spin)
{
    using L = Lay<P>;
    constexpr int NL = L::NL, FAC_LI = L::FAC_LI, FAC_YT = L::FAC_YT, FAC_TI = L::FAC_TI, FACREC = L::FACREC;
    const Ctx c = uniformCtx(cin);
    const RhsSpec sp = uniformSpec(spin);
    const int lane = c.lane, K = c.K;
    const int g = lane >> 4, i = lane & 15;
    const bool scvx = c.ip[IP_SCVX] != 0.;
    // static dual regularisation of the multiplier block, SCvx only (oracle/structured_ipm.hpp: dualReg): there the
    // virtual control really vanishes (E^-1 -> 0) and Theta_0 = E^-1 + Y Y' with rank(M_0) = 3 would turn singular
    const double dual_reg = scvx ? 1e-9 : 0.;
    const XS xs = makeXS<P>(c);
    Tile Z = tileZero(), G = loadRhsW<P>(xs, sp, 0, lane);
    const HsLane hl = hsLane<P>(lane);
    HRaw hcur = loadHRaw<P>(xs, hl, 0, lane);
    for (int k = 0; k < K; k++)
    {
        const FactorRest cur = loadFactorRest<P>(c, xs, sp, k, lane); // arrives during the first elimination below
        HRaw hnxt = hcur;
        if (k + 1 < K)
            hnxt = loadHRaw<P>(xs, hl, k + 1, lane); // prefetch
        double *fk = c.fac + size_t(k) * FACREC;
        double *svk = c.sv + size_t(k) * SVREC;
        Tile Phi = buildHTile<P>(hcur, hl, k, K, lane, scvx);
        if (k > 0)
            Phi = tileAdd(Phi, mm(Z, Z));
        const Tile Li = INVCHOL<NV>(Phi, sh, lane);
        storeTri<NV>(fk + FAC_LI, lane, Li);
        const Tile Lit = transposeTile(Li, sh, lane);
        const Tile a = mm(Lit, G);
        saveCols(svk, sp.n, lane, a);
        if (k == K - 1)
            break;
        const unsigned fm = L::fixedMask(k, K), fmn = L::fixedMask(k + 1, K);
        const Tile Yt = mm(Lit, finishMt(cur.mt, fm, lane));
        storeYt<NL>(fk + FAC_YT, lane, Yt);
        Tile Th = mm(Yt, Yt);
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int row = g + 4 * r;
            if (row == i)
                Th.v[r] = (row < NL) ? Th.v[r] + cur.einv + dual_reg : 1.;
        }
        const Tile Ti = INVCHOL<NL>(Th, sh, lane);
        storeTri<NL>(fk + FAC_TI, lane, Ti);
        const Tile Tit = transposeTile(Ti, sh, lane);
        Z = mm(Tit, finishN<P>(cur.n, fmn, lane));
        const Tile gl = tileSub(rhsLSign(sp, lane, cur.rl), mm(Yt, a));
        const Tile cc = mm(Tit, gl);
        saveCols(svk + NRHS_MAX * 16, sp.n, lane, cc);
        G = tileAdd(cur.rwn, mm(Z, cc));
        hcur = hnxt;
    }
    WAVE_SYNC();
}

struct FwdIn
{
    Tile lit, yt, tit, ti, n, rl, rwn;
};
template <class P>
__device__ inline FwdIn loadFwdIn(const Ctx &c, const XS &xs, const RhsSpec &sp, int k, int lane)
{
    using L = Lay<P>;
    constexpr int NL = L::NL, FAC_LI = L::FAC_LI, FAC_YT = L::FAC_YT, FAC_TI = L::FAC_TI, FACREC = L::FACREC;
    const double *fk = c.fac + size_t(k) * FACREC;
    FwdIn f;
    f.lit = loadTriT<NV>(fk + FAC_LI, lane);
    if (k < c.K - 1)
    {
        f.yt = loadYt<NL>(fk + FAC_YT, lane);
        f.tit = loadTriT<NL>(fk + FAC_TI, lane);
        f.ti = loadTri<NL>(fk + FAC_TI, lane);
        f.n = loadNRaw<P>(c, k, lane);
        f.rl = loadRhsL<P>(c, xs, sp, k, lane);
        f.rwn = loadRhsW<P>(xs, sp, k + 1, lane);
    }
    else
        f.yt = f.tit = f.ti = f.n = f.rl = f.rwn = tileZero();
    return f;
}
template <class P>
SWEEP_FN void fwdSweep(const LDSP Ctx *cin, const RhsSpec &spin)
{
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const RhsSpec sp = uniformSpec(spin);
    const int lane = c.lane, K = c.K;
    const XS xs = makeXS<P>(c);
    Tile G = loadRhsW<P>(xs, sp, 0, lane);
    // two stages of loads in flight: the per-stage MFMA chain (~1k cycles) is much shorter than the loaded-HBM
    // latency, so a distance-1 prefetch still stalls every stage
    FwdIn cur = loadFwdIn<P>(c, xs, sp, 0, lane);
    FwdIn nx1 = K > 1 ? loadFwdIn<P>(c, xs, sp, 1, lane) : cur;
    for (int k = 0; k < K; k++)
    {
        FwdIn nxt = nx1;
        if (k + 2 < K)
            nx1 = loadFwdIn<P>(c, xs, sp, k + 2, lane);
        double *svk = c.sv + size_t(k) * SVREC;
        const Tile a = mm(cur.lit, G);
        saveCols(svk, sp.n, lane, a);
        if (k == K - 1)
            break;
        const Tile gl = tileSub(rhsLSign(sp, lane, cur.rl), mm(cur.yt, a));
        const Tile cc = mm(cur.tit, gl);
        saveCols(svk + NRHS_MAX * 16, sp.n, lane, cc);
        // Z' cc = N' (Ti' cc)
        G = tileAdd(cur.rwn, mm(finishN<P>(cur.n, L::fixedMask(k + 1, K), lane), mm(cur.ti, cc)));
        cur = nxt;
    }
    WAVE_SYNC();
}

template <class P>
__device__ inline void storeSolW(const XS &xs, const RhsSpec &sp, int k, int lane, const Tile &x)
{
    const int g = lane >> 4, i = lane & 15;
    const int kind = colKind(sp, i);
    if (kind)
    {
        const int f = kind == 1 ? int(Lay<P>::X_BCW) : int(Lay<P>::X_VW);
#pragma unroll
        for (int r = 0; r < 4; r++)
            xs.st(xsStage<P>(k), f + g + 4 * r, x.v[r]);
    }
}
template <class P>
__device__ inline void storeSolL(const XS &xs, const RhsSpec &sp, int k, int lane, const Tile &l)
{
    constexpr int NL = Lay<P>::NL;
    const int g = lane >> 4, i = lane & 15;
    const int kind = colKind(sp, i);
    if (kind)
    {
        const int f = kind == 1 ? int(Lay<P>::X_BCL) : int(Lay<P>::X_VL);
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int row = g + 4 * r;
            if (row < NL)
                xs.st(xsStage<P>(k), f + row, l.v[r]);
        }
    }
}

struct BwdIn
{
    Tile nt, tit, ti, y, li, cs, as;
};
template <class P>
__device__ inline BwdIn loadBwdIn(const Ctx &c, const RhsSpec &sp, int k, int lane)
{
    using L = Lay<P>;
    constexpr int NL = L::NL, FAC_LI = L::FAC_LI, FAC_YT = L::FAC_YT, FAC_TI = L::FAC_TI, FACREC = L::FACREC;
    const double *fk = c.fac + size_t(k) * FACREC;
    const double *svk = c.sv + size_t(k) * SVREC;
    BwdIn b;
    b.li = loadTri<NV>(fk + FAC_LI, lane);
    b.as = loadCols(svk, sp.n, lane);
    if (k < c.K - 1)
    {
        b.nt = loadNtRaw<P>(c, k, lane);
        b.tit = loadTriT<NL>(fk + FAC_TI, lane);
        b.ti = loadTri<NL>(fk + FAC_TI, lane);
        b.y = loadYtT<NL>(fk + FAC_YT, lane);
        b.cs = loadCols(svk + NRHS_MAX * 16, sp.n, lane);
    }
    else
        b.nt = b.tit = b.ti = b.y = b.cs = tileZero();
    return b;
}
template <class P>
SWEEP_FN void bwdSweep(const LDSP Ctx *cin, const RhsSpec &spin)
{
    using L = Lay<P>;
    const Ctx c = uniformCtx(cin);
    const RhsSpec sp = uniformSpec(spin);
    const int lane = c.lane, K = c.K;
    const XS xs = makeXS<P>(c);
    BwdIn cur = loadBwdIn<P>(c, sp, K - 1, lane);
    BwdIn nx1 = K > 1 ? loadBwdIn<P>(c, sp, K - 2, lane) : cur;
    Tile x = tileZero();
    for (int k = K - 1; k >= 0; k--)
    {
        BwdIn nxt = nx1;
        if (k > 1)
            nx1 = loadBwdIn<P>(c, sp, k - 2, lane);
        if (k == K - 1)
        {
            x = mm(cur.li, cur.as); // Li' a = L^-T a
        }
        else
        {
            // Z x' - c  with  Z x' = Ti (N x')
            const Tile t = tileSub(mm(cur.tit, mm(finishNt<P>(cur.nt, L::fixedMask(k + 1, K), lane), x)), cur.cs);
            const Tile lam = mm(cur.ti, t);                 // Ti' t = T^-T t
            const Tile s = tileSub(cur.as, mm(cur.y, lam)); // a - Y' lam
            x = mm(cur.li, s);
            storeSolL<P>(xs, sp, k, lane, lam);
        }
        storeSolW<P>(xs, sp, k, lane, x);
        cur = nxt;
    }
    WAVE_SYNC();
}

} // namespace ipm
} // namespace scpp
