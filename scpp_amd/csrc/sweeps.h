// Stage sweeps of the block-tridiagonal KKT system on the tile engine (see tile_engine.h for the algebra).
//   factorSweepFused : factorisation + forward substitution of up to 2 right-hand-side columns (NCV: columns as vectors; 0: as a tile)
//   fwdSweep         : forward substitution only (re-uses the stored tiles)
//   bwdSweep         : backward substitution, writes dw / dlam columns to the exchange records
// Right-hand sides and solutions live in the stage-major exchange records (Lay<P>::X_*).  Column 0 of a 2-column
// sweep is the sigma BORDER column (beta = 0, rho = -S_k, result to X_BCW / X_BCL); the other column reads
// beta from X_BETA (16 entries) and rho from X_RHO (NL entries) and leaves its solution in X_VW / X_VL.
// All sweeps are software-pipelined: the global loads of stage k+1 (k-1) are issued before the dependent
// MFMA / elimination chain of stage k so that HBM/L2 latency overlaps the chain.
// Every global access of a sweep is an UNCONDITIONAL buffer instruction: a lane whose tile entry lies outside the stored
// pattern uses a byte offset beyond the record block, for which the hardware returns 0 (loads) or drops the access (stores).
// With predicated accesses (divergent `if` around a load or store) the compiler cannot count the memory operations between a
// load and its use, and every first use became s_waitcnt vmcnt(0): the prefetch distance of the sweeps existed in the source
// only, and each stage waited for the loads it had just issued (measured in the ISA).
#pragma once
#include "tile_engine.h"

namespace scpp
{
namespace ipm
{

// issue priority of a wavefront inside the factor sweep / the substitution sweeps (s_setprio 0 .. 3; measurement hooks of round 6: two wavefronts
// share a SIMD, and the eliminations are bound by instruction issue while the lane phases wait for memory)
// measured (profiles/r06_ab_priority.json, same box, alternating, all with the non-temporal factor stores): no priorities 5488, factor 1: 5517,
// substitution 1: 5511, factor 1 + substitution 2: 5524 (+0.65 %), factor 2 + substitution 1: 5510
#ifndef IPM_PRIO_FACTOR
#define IPM_PRIO_FACTOR 1
#endif
#ifndef IPM_PRIO_SUBST
#define IPM_PRIO_SUBST 2
#endif
#if defined(SCPP_HIP_EMU)
#define SET_PRIO(p)
#else
#define SET_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
// stages of loads in flight ahead of the dependent chain in the forward / backward sweeps (2 at two waves per SIMD)
#ifndef SWEEP_PREFETCH
#define SWEEP_PREFETCH 2
#endif
#ifndef SWEEPS_INLINE
#define SWEEP_FN static __device__ __attribute__((noinline, disable_tail_calls))
#else
#define SWEEP_FN __device__ inline __attribute__((always_inline))
#endif

// the instance context lives in LDS (one copy per wavefront); out-of-line phases / sweeps re-materialise it in SGPRs
__device__ inline LDSP double *uniformLds(LDSP double *p)
{
#ifdef SCPP_HIP_EMU
    return p;
#else
    return (LDSP double *)(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p);
#endif
}
__device__ inline Ctx uniformCtx(const LDSP Ctx *cin)
{
    Ctx c;
    c.segl = uniformLds(cin->segl);
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
// ---- branch-free buffer access ----
// cache policy per record stream (aux operand of the buffer instructions; gfx950: 1 = sc0, 2 = nt, 16 = sc1).  The factor record and the saved
// forward columns are written once and read back after >= 186 KB of other traffic of the same wavefront (beyond any cache share); the dynamics
// blocks are read once per sweep.  Measurement hooks (round 6, tools/r06_*.sh); 0 = the default policy.
#ifndef IPM_FAC_LD_AUX
#define IPM_FAC_LD_AUX IPM_LD_AUX
#endif
#ifndef IPM_FAC_ST_AUX
#define IPM_FAC_ST_AUX 2 // non-temporal stores of the factor record / saved columns: +0.7 % (same box, alternating, profiles/r06_ab_cache_policy.json);
                         // non-temporal LOADS of them -1.6 %, sc1 stores -2.5 %, non-temporal loads of the dynamics blocks +-0
#endif
#ifndef IPM_DD_LD_AUX
#define IPM_DD_LD_AUX IPM_LD_AUX
#endif
template <int LA, int SA>
struct BufT
{
    __amdgpu_buffer_rsrc_t r;
    __device__ double ld(int vo, int so) const { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, LA)); }
    __device__ void st(int vo, int so, double x) const { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, x), r, vo, so, SA); }
};
typedef BufT<IPM_LD_AUX, IPM_ST_AUX> Buf;
typedef BufT<IPM_FAC_LD_AUX, IPM_FAC_ST_AUX> BufFac;
typedef BufT<IPM_DD_LD_AUX, IPM_ST_AUX> BufDd;
template <class B = Buf>
__device__ inline B makeBuf(const double *p, int ndoubles)
{
    return B{__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, ndoubles * 8, 0x00020000)};
}
// byte offsets of a lane's four tile entries (register r holds row / contraction index g + 4r), VO_OOB where the entry is
// outside the stored pattern; they depend on the lane only and are computed once per sweep
struct Off4
{
    int v[4];
};
template <class B>
__device__ inline Tile ldTile(const B &b, const Off4 &o, int so)
{
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
        t.v[r] = b.ld(o.v[r], so);
    return t;
}
template <class B>
__device__ inline void stTile(const B &b, const Off4 &o, int so, const Tile &t)
{
#pragma unroll
    for (int r = 0; r < 4; r++)
        b.st(o.v[r], so, t.v[r]);
}
template <int n>
__device__ inline Off4 offTri(int lane, int base) // tile[row][col] = L[row][col]
{
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        o.v[r] = (row < n && i <= row) ? (base + triIdx(row, i)) * 8 : VO_OOB;
    }
    return o;
}
template <int n>
__device__ inline Off4 offTriT(int lane, int base) // tile[a][b] = L[b][a]
{
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int a = g + 4 * r;
        o.v[r] = (i < n && a <= i) ? (base + triIdx(i, a)) * 8 : VO_OOB;
    }
    return o;
}
// the identity outside the leading n x n block (loads returned 0 there)
template <int n>
__device__ inline Tile finishTri(const Tile &raw, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Tile t = raw;
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (4 * r + 3 >= n) // this register holds a row >= n
            t.v[r] = (g + 4 * r >= n && g + 4 * r == i) ? 1. : raw.v[r];
    return t;
}
template <int NL>
__device__ inline Off4 offYt(int lane, int base) // tile[a][b] = Yt[a][b], row-major with pitch NL
{
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
        o.v[r] = i < NL ? (base + (g + 4 * r) * NL + i) * 8 : VO_OOB;
    return o;
}
template <int NL>
__device__ inline Off4 offYtT(int lane, int base) // tile[a][b] = Yt[b][a]
{
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int a = g + 4 * r;
        o.v[r] = a < NL ? (base + i * NL + a) * 8 : VO_OOB;
    }
    return o;
}
// saved right-hand-side columns (a, c of the forward pass): column i < n at p[i * 16 + row]
__device__ inline Off4 offCols(int lane, int n, int base)
{
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
        o.v[r] = i < n ? (base + i * 16 + g + 4 * r) * 8 : VO_OOB;
    return o;
}

// what a sweep reads and writes: the instance's record blocks as buffer resources + the byte offset of a stage in each
template <class P>
struct SweepIO
{
    using L = Lay<P>;
    BufFac fac, sv;
    Buf sx;
    BufDd A, B, C;
    __device__ explicit SweepIO(const Ctx &c)
        : fac(makeBuf<BufFac>(c.fac, c.K * L::FACREC)), sv(makeBuf<BufFac>(c.sv, c.K * SVREC)), sx(makeBuf<Buf>(c.sx, c.K * L::XREC)),
          A(makeBuf<BufDd>(c.A, (c.K - 1) * P::NX * P::NX)), B(makeBuf<BufDd>(c.B, (c.K - 1) * P::NX * P::NU)),
          C(makeBuf<BufDd>(c.C, (c.K - 1) * P::NX * P::NU))
    {
    }
    static __device__ int sFac(int k) { return k * (L::FACREC * 8); }
    static __device__ int sSv(int k) { return k * (SVREC * 8); }
    static __device__ int sX(int k) { return k * (L::XREC * 8); }
    static __device__ int sA(int k) { return k * (P::NX * P::NX * 8); }
    static __device__ int sBC(int k) { return k * (P::NX * P::NU * 8); }
};

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

__device__ inline int colKind(const RhsSpec &sp, int i) // 0 none, 1 border, 2 regular
{
    if (sp.n == 2)
        return i == 0 ? 1 : (i == 1 ? 2 : 0);
    return i == 0 ? 2 : 0;
}

// right-hand sides / solutions in the exchange record: the border column reads -S_k (X_S) and writes X_BCW / X_BCL
template <class P>
__device__ inline Off4 offRhsW(const RhsSpec &sp, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
        o.v[r] = colKind(sp, i) == 2 ? (Lay<P>::X_BETA + g + 4 * r) * 8 : VO_OOB;
    return o;
}
template <class P>
__device__ inline Off4 offRhsL(const RhsSpec &sp, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    const int kind = colKind(sp, i);
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        o.v[r] = (kind && row < Lay<P>::NL) ? ((kind == 1 ? Lay<P>::X_S : Lay<P>::X_RHO) + row) * 8 : VO_OOB;
    }
    return o;
}
template <class P>
__device__ inline Off4 offSolW(const RhsSpec &sp, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    const int kind = colKind(sp, i);
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
        o.v[r] = kind ? ((kind == 1 ? Lay<P>::X_BCW : Lay<P>::X_VW) + g + 4 * r) * 8 : VO_OOB;
    return o;
}
template <class P>
__device__ inline Off4 offSolL(const RhsSpec &sp, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    const int kind = colKind(sp, i);
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        o.v[r] = (kind && row < Lay<P>::NL) ? ((kind == 1 ? Lay<P>::X_BCL : Lay<P>::X_VL) + row) * 8 : VO_OOB;
    }
    return o;
}
// border column stores S: the sign is applied at use so that the load itself carries no arithmetic
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

// ---- right-hand-side COLUMNS in V layout (vector sweeps; NC = 1: one regular column, NC = 2: sigma border + regular column) ----
// column c of an NC-column sweep: border (reads -S_k, beta = 0, writes X_BCW / X_BCL) or regular (X_BETA / X_RHO -> X_VW / X_VL)
template <int NC>
__device__ inline constexpr bool colIsBorder(int c)
{
    return NC == 2 && c == 0;
}
template <class P, int NC>
struct VCols
{
    using L = Lay<P>;
    int rw[NC], rl[NC], solW[NC], solL[NC], colL[NC], colS[NC];
    __device__ explicit VCols(int lane)
    {
        const int e = vElem(lane);
        const bool st0 = (lane & 3) == 0; // every element lives in four lanes: one of them stores
#pragma unroll
        for (int c = 0; c < NC; c++)
        {
            const bool border = colIsBorder<NC>(c);
            rw[c] = border ? VO_OOB : (L::X_BETA + e) * 8;
            rl[c] = e < L::NL ? ((border ? L::X_S : L::X_RHO) + e) * 8 : VO_OOB;
            solW[c] = st0 ? ((border ? L::X_BCW : L::X_VW) + e) * 8 : VO_OOB;
            solL[c] = (st0 && e < L::NL) ? ((border ? L::X_BCL : L::X_VL) + e) * 8 : VO_OOB;
            colL[c] = (c * 16 + e) * 8; // saved forward intermediates: column c at [c * 16 + row] (the layout of offCols)
            colS[c] = st0 ? (c * 16 + e) * 8 : VO_OOB;
        }
    }
    // the border column stores S: the sign is applied at use
    __device__ static double rlSign(int c, double v) { return colIsBorder<NC>(c) ? -v : v; }
};

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
    Off4 off; // byte offsets in the exchange record (VO_OOB outside the pattern: the load returns 0)
};
template <class P>
__device__ inline HsLane hsLane(int lane)
{
    const int g = lane >> 4, i = lane & 15;
    HsLane h;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        h.idx[r] = hsIndex<P>(g + 4 * r, i);
        h.off.v[r] = h.idx[r] >= 0 ? (Lay<P>::X_HS + h.idx[r]) * 8 : VO_OOB;
    }
    return h;
}
template <class P>
__device__ inline HRaw loadHRaw(const SweepIO<P> &io, const HsLane &hl, int k, int lane)
{
    using L = Lay<P>;
    const int g = lane >> 4, i = lane & 15;
    const int sk = io.sX(k);
    HRaw h;
    h.e2 = io.sx.ld(L::X_HC * 8, sk);
    h.cc = io.sx.ld((L::X_HC + 1) * 8, sk);
    h.wcol = io.sx.ld((L::X_WBT + i) * 8, sk);
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        h.wrow[r] = io.sx.ld((L::X_WBT + g + 4 * r) * 8, sk);
        h.hs[r] = io.sx.ld(hl.off.v[r], sk);
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
        v += h.hs[r]; // 0 outside the pattern
        if ((fm & (1u << row)) || (fm & (1u << i)))
            v = (row == i) ? 1. : 0.;
        t.v[r] = v;
    }
    return t;
}
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
// raw entries of [A|B] / C arranged for the M' and N tiles (mask and sign applied at use)
// M' tile: entry (var j = g+4r, dyn row i); the A part and the B part come from different arrays
template <class P>
struct OffMt
{
    Off4 a, b;
};
template <class P>
__device__ inline OffMt<P> offMt(int lane)
{
    using L = Lay<P>;
    const int g = lane >> 4, i = lane & 15;
    OffMt<P> o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int j = g + 4 * r;
        o.a.v[r] = (i < L::NL && j < P::NXV) ? (i * P::NX + varColumn<P>(j)) * 8 : VO_OOB;
        o.b.v[r] = (i < L::NL && j >= P::NXV && j < L::NVU) ? (i * P::NU + varColumn<P>(j)) * 8 : VO_OOB;
    }
    return o;
}
template <class P>
__device__ inline Tile loadMtRaw(const SweepIO<P> &io, const OffMt<P> &o, int k)
{
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        // register r holds variables 4r .. 4r+3: all states, all inputs (or unused), or both
        if (4 * r + 3 < P::NXV)
            t.v[r] = io.A.ld(o.a.v[r], io.sA(k));
        else if (4 * r >= P::NXV)
            t.v[r] = (4 * r < Lay<P>::NVU) ? io.B.ld(o.b.v[r], io.sBC(k)) : 0.;
        else
            t.v[r] = io.A.ld(o.a.v[r], io.sA(k)) + io.B.ld(o.b.v[r], io.sBC(k)); // one of the two is out of range -> 0
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
__device__ inline Off4 offN(int lane) // entry (dyn row g+4r, var i): only the C part is loaded
{
    using L = Lay<P>;
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        o.v[r] = (i >= P::NXV && i < L::NVU && row < L::NL) ? (row * P::NU + varColumn<P>(i)) * 8 : VO_OOB;
    }
    return o;
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
__device__ inline Off4 offNt(int lane)
{
    using L = Lay<P>;
    const int g = lane >> 4, i = lane & 15;
    Off4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int a = g + 4 * r;
        o.v[r] = (i < L::NL && a >= P::NXV && a < L::NVU) ? (i * P::NU + varColumn<P>(a)) * 8 : VO_OOB;
    }
    return o;
}
template <class P>
__device__ inline Tile loadNtRaw(const SweepIO<P> &io, const Off4 &o, int k)
{
    Tile t = tileZero();
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        if (4 * r + 3 < P::NXV || 4 * r >= Lay<P>::NVU) // this register holds no input variable (RocketQuat: inputs 13..15 live in register 3)
            continue;
        t.v[r] = io.C.ld(o.v[r], io.sBC(k));
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
struct FactorOffs
{
    OffMt<P> mt;
    Off4 n, rl, rw, cols, triV, triL, triVT, triLT, yt;
    int einv;
};

// NCV > 0: the NCV right-hand-side columns (1: SCvx mode, no sigma border; 2: [border | column]) in VECTOR form -- the forward substitution
// fused into the factorisation then runs on vectors (tile_engine.h: mv, four 4-pass matrix-core instructions per product instead of four
// 16-pass ones) and a column, its intermediates and the stage's right-hand sides are single registers instead of tiles.  NCV == 0: tiles.
template <class P, int NCV>
SWEEP_FN void factorSweepFused(const LDSP Ctx *cin, TileShared &sh, const RhsSpec &spin)
{
    EMU_PHASE("factorSweepFused");
    using L = Lay<P>;
    constexpr int NL = L::NL;
    const Ctx c = uniformCtx(cin);
    const RhsSpec sp = uniformSpec(spin);
    const int lane = c.lane, K = c.K;
    const int g = lane >> 4, i = lane & 15;
    const bool scvx = c.ip[IP_SCVX] != 0.;
    if (IPM_PRIO_FACTOR != IPM_PRIO_LANE)
        SET_PRIO(IPM_PRIO_FACTOR);
    // static dual regularisation of the multiplier block, SCvx only (oracle/structured_ipm.hpp: dualReg): there the
    // virtual control really vanishes (E^-1 -> 0) and Theta_0 = E^-1 + Y Y' with rank(M_0) = 3 would turn singular
    const double dual_reg = scvx ? 1e-9 : 0.;
    const SweepIO<P> io(c);
    FactorOffs<P> o;
    o.mt = offMt<P>(lane);
    o.n = offN<P>(lane);
    o.rl = offRhsL<P>(sp, lane);
    o.rw = offRhsW<P>(sp, lane);
    o.cols = offCols(lane, sp.n, 0);
    o.triV = offTri<NV>(lane, L::FAC_LI);
    o.triL = offTri<NL>(lane, L::FAC_TI);
    o.triVT = offTriT<NV>(lane, L::FAC_LI);
    o.triLT = offTriT<NL>(lane, L::FAC_TI);
    o.yt = offYt<NL>(lane, L::FAC_YT);
    o.einv = i < NL ? (L::X_EINV + i) * 8 : VO_OOB;
#ifdef IPM_PROFILE
    double pf0 = 0., pf1 = 0., pfA = 0., pfC = 0., pfT = 0., pfL = 0., pfH = 0., pfZ = 0.;
    const long long tfs = clock64();
    long long tstage = tfs;
#endif
    // vector form of the right-hand-side columns (VEC): V-layout offsets, one of the four lanes of an element stores
    constexpr bool VEC = NCV > 0;
    constexpr int NC = VEC ? NCV : 1;
    const VCols<P, NC> vc(lane);
    Tile Z = tileZero(), G = VEC ? tileZero() : ldTile(io.sx, o.rw, io.sX(0));
    double Gv[NC];
#pragma unroll
    for (int q = 0; q < NC; q++)
        Gv[q] = VEC ? io.sx.ld(vc.rw[q], io.sX(0)) : 0.;
    const HsLane hl = hsLane<P>(lane);
    // One stage.  What is consumed right after the first elimination (M' and E^-1: 12 VGPRs) and the Hessian entries are
    // requested ONE STAGE AHEAD into buffers that rotate by name (stage loop unrolled two-fold) -- under this traffic a load
    // takes about as long as the first elimination, which left a wait in front of Yt = Li M'; the tiles of the stage's tail (N,
    // right-hand sides: 24 VGPRs) are requested at the start of their own stage (prefetching them as well spills).
    struct Early
    {
        Tile mt;
        double einv;
        HRaw h;
    };
    auto loadEarly = [&](int k) {
        const int ks = k < K - 1 ? k : K - 2; // the last stage has no segment: it re-reads segment K-2 (in range, unused)
        Early e;
        e.mt = loadMtRaw<P>(io, o.mt, ks);
        e.einv = io.sx.ld(o.einv, io.sX(ks));
        e.h = loadHRaw<P>(io, hl, k, lane);
        return e;
    };
    auto stage = [&](int k, const Early &cur, Early &nxt) -> bool {
        const int ks = k < K - 1 ? k : K - 2, kn = k + 1 < K ? k + 1 : k;
        const Tile cn = ldTile(io.C, o.n, io.sBC(ks));
        Tile crl, crwn;
        double vrl[NC], vrwn[NC];
        if constexpr (VEC)
        {
#pragma unroll
            for (int q = 0; q < NC; q++)
            {
                vrl[q] = io.sx.ld(vc.rl[q], io.sX(ks));
                vrwn[q] = io.sx.ld(vc.rw[q], io.sX(ks + 1));
            }
        }
        else
        {
            crl = ldTile(io.sx, o.rl, io.sX(ks));
            crwn = ldTile(io.sx, o.rw, io.sX(ks + 1));
        }
        nxt = loadEarly(kn);
        LOADS_ISSUED();
#ifdef IPM_PROFILE
        const long long tfl = clock64();
        pfL += double(tfl - tstage); // (tstage was advanced to the end of the previous stage's tail below)
#endif
        Tile Phi = buildHTile<P>(cur.h, hl, k, K, lane, scvx);
#ifdef IPM_PROFILE
        const long long tfh = clock64();
        pfH += double(tfh - tfl);
#endif
        if (k > 0)
            Phi = tileAdd(Phi, mm(Z, Z));
#ifdef IPM_PROFILE
        const long long tf0 = clock64();
        pfZ += double(tf0 - tfh);
        pfA += double(tf0 - tstage); // stage head: loads issued, H tile, Z'Z (+ pfT: the tail of the previous stage)
#endif
#if INVCHOL_TRANSPOSED
        const Tile Lit = invCholFactorT<NV>(Phi, lane);
#ifdef IPM_PROFILE
        const long long tf1 = clock64();
        pf0 += double(tf1 - tf0);
#endif
        stTile(io.fac, o.triVT, io.sFac(k), Lit); // (same record: the transposed offsets store L[b][a] = Lit[a][b])
#else
        const Tile Li = INVCHOL<NV>(Phi, sh, lane);
#ifdef IPM_PROFILE
        const long long tf1 = clock64();
        pf0 += double(tf1 - tf0);
#endif
        stTile(io.fac, o.triV, io.sFac(k), Li);
        const Tile Lit = transposeTile(Li, sh, lane);
#endif
        Tile a;
        double va[NC];
        if constexpr (VEC)
        {
#pragma unroll
            for (int q = 0; q < NC; q++)
            {
                va[q] = mv(Lit, Gv[q]); // Li g
                io.sv.st(vc.colS[q], io.sSv(k), va[q]);
            }
        }
        else
        {
            a = mm(Lit, G);
            stTile(io.sv, o.cols, io.sSv(k), a);
        }
        if (k == K - 1)
            return false;
        const unsigned fm = L::fixedMask(k, K), fmn = L::fixedMask(k + 1, K);
        const Tile Yt = mm(Lit, finishMt(cur.mt, fm, lane));
        stTile(io.fac, o.yt, io.sFac(k), Yt);
        Tile Th = mm(Yt, Yt);
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int row = g + 4 * r;
            if (row == i)
                Th.v[r] = (row < NL) ? Th.v[r] + cur.einv + dual_reg : 1.;
        }
#ifdef IPM_PROFILE
        const long long tf2 = clock64();
        pfC += double(tf2 - tf1); // between the eliminations: store Li, transpose, a, Yt, Theta
#endif
#if INVCHOL_TRANSPOSED
        const Tile Tit = invCholFactorT<NL>(Th, lane);
#ifdef IPM_PROFILE
        tstage = clock64();
        pf1 += double(tstage - tf2);
#endif
        stTile(io.fac, o.triLT, io.sFac(k), Tit);
#else
        const Tile Ti = INVCHOL<NL>(Th, sh, lane);
#ifdef IPM_PROFILE
        tstage = clock64();
        pf1 += double(tstage - tf2);
#endif
        stTile(io.fac, o.triL, io.sFac(k), Ti);
        const Tile Tit = transposeTile(Ti, sh, lane);
#endif
        Z = mm(Tit, finishN<P>(cn, fmn, lane));
        if constexpr (VEC)
        {
#pragma unroll
            for (int q = 0; q < NC; q++)
            {
                const double vgl = VCols<P, NC>::rlSign(q, vrl[q]) - mv(Yt, va[q]); // rho - Yt' a
                const double vcc = mv(Tit, vgl);                                    // Ti gl
                io.sv.st(vc.colS[q], io.sSv(k) + NRHS_MAX * 16 * 8, vcc);
                Gv[q] = vrwn[q] + mv(Z, vcc); // beta' + Z' c
            }
        }
        else
        {
            const Tile gl = tileSub(rhsLSign(sp, lane, crl), mm(Yt, a));
            const Tile cc = mm(Tit, gl);
            stTile(io.sv, o.cols, io.sSv(k) + NRHS_MAX * 16 * 8, cc);
            G = tileAdd(crwn, mm(Z, cc));
        }
#ifdef IPM_PROFILE
        {
            const long long tft = clock64();
            pfT += double(tft - tstage); // tail: store Ti, transpose, Z, forward pass of the columns
            pfA += double(tft - tstage);
            tstage = tft;
        }
#endif
        return true;
    };
    Early e0 = loadEarly(0), e1;
    for (int k = 0; k < K; k += 2)
    {
        if (!stage(k, e0, e1))
            break;
        if (!stage(k + 1, e1, e0))
            break;
    }
#ifdef IPM_PROFILE
    if (lane == 0)
    {
        sh.prof[0] += pf0;
        sh.prof[1] += pf1;
        sh.prof[4] += pfA;
        sh.prof[5] += pfC;
        sh.prof[6] += pfT;
        sh.prof[7] += pfL;
        sh.prof[8] += pfH;
        sh.prof[9] += pfZ;
        sh.prof[2] += double(clock64() - tfs);
        sh.prof[3] += 1.;
    }
#endif
    if (IPM_PRIO_FACTOR != IPM_PRIO_LANE)
        SET_PRIO(IPM_PRIO_LANE);
    WAVE_SYNC();
}

struct FwdIn
{
    Tile lit, yt, tit, ti, n, rl, rwn;
};
template <class P>
struct FwdOffs
{
    Off4 lit, yt, tit, ti, n, rl, rw, cols;
};
template <class P>
__device__ inline FwdIn loadFwdIn(const SweepIO<P> &io, const FwdOffs<P> &o, int k, int K)
{
    // the last stage has no segment: it re-reads segment K-2 (in range, unused)
    const int ks = k < K - 1 ? k : K - 2;
    FwdIn f;
    f.lit = ldTile(io.fac, o.lit, io.sFac(k));
    f.yt = ldTile(io.fac, o.yt, io.sFac(ks));
    f.tit = ldTile(io.fac, o.tit, io.sFac(ks));
    f.ti = ldTile(io.fac, o.ti, io.sFac(ks));
    f.n = ldTile(io.C, o.n, io.sBC(ks));
    f.rl = ldTile(io.sx, o.rl, io.sX(ks));
    f.rwn = ldTile(io.sx, o.rw, io.sX(ks + 1));
    return f;
}
template <class P>
SWEEP_FN void fwdSweep(const LDSP Ctx *cin, const RhsSpec &spin)
{
    EMU_PHASE("fwdSweep");
    using L = Lay<P>;
    constexpr int NL = L::NL;
    const Ctx c = uniformCtx(cin);
    const RhsSpec sp = uniformSpec(spin);
    const int lane = c.lane, K = c.K;
    const SweepIO<P> io(c);
    FwdOffs<P> o;
    o.lit = offTriT<NV>(lane, L::FAC_LI);
    o.yt = offYt<NL>(lane, L::FAC_YT);
    o.tit = offTriT<NL>(lane, L::FAC_TI);
    o.ti = offTri<NL>(lane, L::FAC_TI);
    o.n = offN<P>(lane);
    o.rl = offRhsL<P>(sp, lane);
    o.rw = offRhsW<P>(sp, lane);
    o.cols = offCols(lane, sp.n, 0);
    Tile G = ldTile(io.sx, o.rw, io.sX(0));
    // Three prefetch buffers rotated BY NAME (stage loop unrolled three-fold): loads of stage k+2 are in flight while stage k
    // computes.  Rotating them by register copies (cur = nxt; nxt = nx1) makes every copy wait for the load it copies --
    // i.e. for the newest loads -- now that the waits are counted exactly (see the header), which was one exposed memory round
    // trip per stage.
    auto stage = [&](int k, const FwdIn &cur) -> bool {
        const Tile a = mm(finishTri<NV>(cur.lit, lane), G);
        stTile(io.sv, o.cols, io.sSv(k), a);
        if (k == K - 1)
            return false;
        const Tile gl = tileSub(rhsLSign(sp, lane, cur.rl), mm(cur.yt, a));
        const Tile cc = mm(finishTri<NL>(cur.tit, lane), gl);
        stTile(io.sv, o.cols, io.sSv(k) + NRHS_MAX * 16 * 8, cc);
        // Z' cc = N' (Ti' cc)
        G = tileAdd(cur.rwn, mm(finishN<P>(cur.n, L::fixedMask(k + 1, K), lane), mm(finishTri<NL>(cur.ti, lane), cc)));
        return true;
    };
    auto clampK = [&](int k) { return k < K ? k : K - 1; };
#if SWEEP_PREFETCH == 1
    FwdIn b0 = loadFwdIn<P>(io, o, 0, K), b1;
    for (int k = 0; k < K; k += 2)
    {
        b1 = loadFwdIn<P>(io, o, clampK(k + 1), K);
        LOADS_ISSUED();
        if (!stage(k, b0))
            break;
        b0 = loadFwdIn<P>(io, o, clampK(k + 2), K);
        LOADS_ISSUED();
        if (!stage(k + 1, b1))
            break;
    }
#else
    FwdIn b0 = loadFwdIn<P>(io, o, 0, K), b1 = loadFwdIn<P>(io, o, clampK(1), K), b2;
    for (int k = 0; k < K; k += 3)
    {
        b2 = loadFwdIn<P>(io, o, clampK(k + 2), K);
        LOADS_ISSUED();
        if (!stage(k, b0))
            break;
        b0 = loadFwdIn<P>(io, o, clampK(k + 3), K);
        LOADS_ISSUED();
        if (!stage(k + 1, b1))
            break;
        b1 = loadFwdIn<P>(io, o, clampK(k + 4), K);
        LOADS_ISSUED();
        if (!stage(k + 2, b2))
            break;
    }
#endif
    WAVE_SYNC();
}

struct BwdIn
{
    Tile nt, tit, ti, y, li, cs, as;
};
template <class P>
struct BwdOffs
{
    Off4 nt, tit, ti, y, li, cols, solW, solL;
};
template <class P>
__device__ inline BwdIn loadBwdIn(const SweepIO<P> &io, const BwdOffs<P> &o, int k, int K)
{
    const int ks = k < K - 1 ? k : K - 2;
    BwdIn b;
    b.li = ldTile(io.fac, o.li, io.sFac(k));
    b.as = ldTile(io.sv, o.cols, io.sSv(k));
    b.nt = loadNtRaw<P>(io, o.nt, ks);
    b.tit = ldTile(io.fac, o.tit, io.sFac(ks));
    b.ti = ldTile(io.fac, o.ti, io.sFac(ks));
    b.y = ldTile(io.fac, o.y, io.sFac(ks));
    b.cs = ldTile(io.sv, o.cols, io.sSv(ks) + NRHS_MAX * 16 * 8);
    return b;
}
template <class P>
SWEEP_FN void bwdSweep(const LDSP Ctx *cin, const RhsSpec &spin)
{
    EMU_PHASE("bwdSweep");
    using L = Lay<P>;
    constexpr int NL = L::NL;
    const Ctx c = uniformCtx(cin);
    const RhsSpec sp = uniformSpec(spin);
    const int lane = c.lane, K = c.K;
    const SweepIO<P> io(c);
    BwdOffs<P> o;
    o.nt = offNt<P>(lane);
    o.tit = offTriT<NL>(lane, L::FAC_TI);
    o.ti = offTri<NL>(lane, L::FAC_TI);
    o.y = offYtT<NL>(lane, L::FAC_YT);
    o.li = offTri<NV>(lane, L::FAC_LI);
    o.cols = offCols(lane, sp.n, 0);
    o.solW = offSolW<P>(sp, lane);
    o.solL = offSolL<P>(sp, lane);
    Tile x = tileZero();
    // three prefetch buffers rotated by name, see fwdSweep
    auto stage = [&](int k, const BwdIn &cur) {
        if (k == K - 1)
        {
            x = mm(finishTri<NV>(cur.li, lane), cur.as); // Li' a = L^-T a
        }
        else
        {
            // Z x' - c  with  Z x' = Ti (N x')
            const Tile t = tileSub(mm(finishTri<NL>(cur.tit, lane), mm(finishNt<P>(cur.nt, L::fixedMask(k + 1, K), lane), x)), cur.cs);
            const Tile lam = mm(finishTri<NL>(cur.ti, lane), t); // Ti' t = T^-T t
            const Tile s = tileSub(cur.as, mm(cur.y, lam));      // a - Y' lam
            x = mm(finishTri<NV>(cur.li, lane), s);
            stTile(io.sx, o.solL, io.sX(k), lam);
        }
        stTile(io.sx, o.solW, io.sX(k), x);
    };
    auto clampK = [&](int k) { return k > 0 ? k : 0; };
#if SWEEP_PREFETCH == 1
    BwdIn b0 = loadBwdIn<P>(io, o, K - 1, K), b1;
    for (int k = K - 1; k >= 0; k -= 2)
    {
        b1 = loadBwdIn<P>(io, o, clampK(k - 1), K);
        LOADS_ISSUED();
        stage(k, b0);
        if (k - 1 < 0)
            break;
        b0 = loadBwdIn<P>(io, o, clampK(k - 2), K);
        LOADS_ISSUED();
        stage(k - 1, b1);
    }
#else
    BwdIn b0 = loadBwdIn<P>(io, o, K - 1, K), b1 = loadBwdIn<P>(io, o, clampK(K - 2), K), b2;
    for (int k = K - 1; k >= 0; k -= 3)
    {
        b2 = loadBwdIn<P>(io, o, clampK(k - 2), K);
        LOADS_ISSUED();
        stage(k, b0);
        if (k - 1 < 0)
            break;
        b0 = loadBwdIn<P>(io, o, clampK(k - 3), K);
        LOADS_ISSUED();
        stage(k - 1, b1);
        if (k - 2 < 0)
            break;
        b1 = loadBwdIn<P>(io, o, clampK(k - 4), K);
        LOADS_ISSUED();
        stage(k - 2, b2);
    }
#endif
    WAVE_SYNC();
}

// =====================================================================================================================
// Substitution sweeps on the matrix-VECTOR engine (tile_engine.h: mv; round 4), one or two right-hand-side columns.
// fwdSweep / bwdSweep above carry up to two right-hand-side columns through X'Y products of 16 x 16 tiles: four 16-pass matrix-core
// instructions per product whatever the number of columns.  Every sweep of the SCvx mode (no sigma border, specSingle) and the corrector
// sweeps of the SC mode have ONE column, the predictor sweeps of the SC mode two ([sigma border | column]): here the same recursion runs
// on vectors in V layout and tiles loaded in A4 layout, four 4-pass v_mfma_f64_4x4x4_4b_f64 + four DPP row broadcasts per product and
// column, and a right-hand side is one register instead of a tile.  Same operands, same records, and the contraction is grouped in the
// same blocks of four as in the 16-wide instruction: results are BITWISE those of the tile sweeps (tests/test_emu_kernels.py compiles
// both; tests/tools/lib_equal.py, sc_mode_ab.py on the GPU).  The factorisation itself keeps the tile engine: its products are matrix x
// matrix; only its fused forward substitution is vectors.
// =====================================================================================================================
// mv(X, v) = X' v for a tile X in the D layout of the 16-wide instruction: "A4 layout of T" (tile_engine.h) IS the D layout of T', so the
// vector sweeps load exactly the tiles -- same offsets, same coalescing, same finish* masks -- the tile sweeps above load, and replace
// mm(X, right-hand-side tile) by mv(X, right-hand-side vector).
// (prefetch distance: two stages, three buffers rotated by name; four buffers = three stages ahead measured 5116 against 5137 converged/s,
//  round 4: the vector sweeps are bound by their chain of dependent products, not by memory latency)
template <int NC>
struct FwdVIn
{
    Tile lit, yt, tit, ti, n;
    double rl[NC], rwn[NC];
};
template <class P, int NC>
SWEEP_FN void fwdSweepV(const LDSP Ctx *cin)
{
    EMU_PHASE("fwdSweep");
    using L = Lay<P>;
    constexpr int NL = L::NL;
    const Ctx c = uniformCtx(cin);
    const int lane = c.lane, K = c.K;
    const SweepIO<P> io(c);
    if (IPM_PRIO_SUBST != IPM_PRIO_LANE)
        SET_PRIO(IPM_PRIO_SUBST);
    const Off4 oLit = offTriT<NV>(lane, L::FAC_LI), oYt = offYt<NL>(lane, L::FAC_YT), oTit = offTriT<NL>(lane, L::FAC_TI), oTi = offTri<NL>(lane, L::FAC_TI),
               oN = offN<P>(lane);
    const VCols<P, NC> vc(lane);
    auto load = [&](int k) {
        const int ks = k < K - 1 ? k : K - 2; // the last stage has no segment: it re-reads segment K-2 (in range, unused)
        FwdVIn<NC> f;
        f.lit = ldTile(io.fac, oLit, io.sFac(k));
        f.yt = ldTile(io.fac, oYt, io.sFac(ks));
        f.tit = ldTile(io.fac, oTit, io.sFac(ks));
        f.ti = ldTile(io.fac, oTi, io.sFac(ks));
        f.n = ldTile(io.C, oN, io.sBC(ks));
#pragma unroll
        for (int q = 0; q < NC; q++)
        {
            f.rl[q] = io.sx.ld(vc.rl[q], io.sX(ks));
            f.rwn[q] = io.sx.ld(vc.rw[q], io.sX(ks + 1));
        }
        return f;
    };
    double G[NC];
#pragma unroll
    for (int q = 0; q < NC; q++)
        G[q] = io.sx.ld(vc.rw[q], io.sX(0));
    auto stage = [&](int k, const FwdVIn<NC> &cur) -> bool {
        double a[NC];
#pragma unroll
        for (int q = 0; q < NC; q++)
        {
            a[q] = mv(cur.lit, G[q]); // Li g
            io.sv.st(vc.colS[q], io.sSv(k), a[q]);
        }
        if (k == K - 1)
            return false;
#pragma unroll
        for (int q = 0; q < NC; q++)
        {
            const double gl = VCols<P, NC>::rlSign(q, cur.rl[q]) - mv(cur.yt, a[q]); // rho - Yt' a
            const double cc = mv(cur.tit, gl);                                       // Ti gl
            io.sv.st(vc.colS[q], io.sSv(k) + NRHS_MAX * 16 * 8, cc);
            G[q] = cur.rwn[q] + mv(finishN<P>(cur.n, L::fixedMask(k + 1, K), lane), mv(cur.ti, cc)); // beta' + N' (Ti' c)
        }
        return true;
    };
    auto clampK = [&](int k) { return k < K ? k : K - 1; };
    FwdVIn<NC> b0 = load(0), b1 = load(clampK(1)), b2;
    for (int k = 0; k < K; k += 3)
    {
        b2 = load(clampK(k + 2));
        LOADS_ISSUED();
        if (!stage(k, b0))
            break;
        b0 = load(clampK(k + 3));
        LOADS_ISSUED();
        if (!stage(k + 1, b1))
            break;
        b1 = load(clampK(k + 4));
        LOADS_ISSUED();
        if (!stage(k + 2, b2))
            break;
    }
    if (IPM_PRIO_SUBST != IPM_PRIO_LANE)
        SET_PRIO(IPM_PRIO_LANE);
    WAVE_SYNC();
}

template <int NC>
struct BwdVIn
{
    Tile li, nt, tit, ti, y;
    double as[NC], cs[NC];
};
template <class P, int NC>
SWEEP_FN void bwdSweepV(const LDSP Ctx *cin)
{
    EMU_PHASE("bwdSweep");
    using L = Lay<P>;
    constexpr int NL = L::NL;
    const Ctx c = uniformCtx(cin);
    const int lane = c.lane, K = c.K;
    const SweepIO<P> io(c);
    if (IPM_PRIO_SUBST != IPM_PRIO_LANE)
        SET_PRIO(IPM_PRIO_SUBST);
    const Off4 oLi = offTri<NV>(lane, L::FAC_LI), oNt = offNt<P>(lane), oTit = offTriT<NL>(lane, L::FAC_TI), oTi = offTri<NL>(lane, L::FAC_TI),
               oY = offYtT<NL>(lane, L::FAC_YT);
    const VCols<P, NC> vc(lane);
    auto load = [&](int k) {
        const int ks = k < K - 1 ? k : K - 2;
        BwdVIn<NC> b;
        b.li = ldTile(io.fac, oLi, io.sFac(k));
#pragma unroll
        for (int q = 0; q < NC; q++)
            b.as[q] = io.sv.ld(vc.colL[q], io.sSv(k));
        b.nt = loadNtRaw<P>(io, oNt, ks);
        b.tit = ldTile(io.fac, oTit, io.sFac(ks));
        b.ti = ldTile(io.fac, oTi, io.sFac(ks));
        b.y = ldTile(io.fac, oY, io.sFac(ks));
#pragma unroll
        for (int q = 0; q < NC; q++)
            b.cs[q] = io.sv.ld(vc.colL[q], io.sSv(ks) + NRHS_MAX * 16 * 8);
        return b;
    };
    double x[NC];
#pragma unroll
    for (int q = 0; q < NC; q++)
        x[q] = 0.;
    auto stage = [&](int k, const BwdVIn<NC> &cur) {
        if (k == K - 1)
        {
#pragma unroll
            for (int q = 0; q < NC; q++)
                x[q] = mv(cur.li, cur.as[q]); // Li' a
        }
        else
        {
#pragma unroll
            for (int q = 0; q < NC; q++)
            {
                const double t = mv(cur.tit, mv(finishNt<P>(cur.nt, L::fixedMask(k + 1, K), lane), x[q])) - cur.cs[q]; // Ti (N x') - c
                const double lam = mv(cur.ti, t);                          // Ti' t
                const double s = cur.as[q] - mv(cur.y, lam);               // a - Yt lam
                x[q] = mv(cur.li, s);
                io.sx.st(vc.solL[q], io.sX(k), lam);
            }
        }
#pragma unroll
        for (int q = 0; q < NC; q++)
            io.sx.st(vc.solW[q], io.sX(k), x[q]);
    };
    auto clampK = [&](int k) { return k > 0 ? k : 0; };
    BwdVIn<NC> b0 = load(K - 1), b1 = load(clampK(K - 2)), b2;
    for (int k = K - 1; k >= 0; k -= 3)
    {
        b2 = load(clampK(k - 2));
        LOADS_ISSUED();
        stage(k, b0);
        if (k - 1 < 0)
            break;
        b0 = load(clampK(k - 3));
        LOADS_ISSUED();
        stage(k - 1, b1);
        if (k - 2 < 0)
            break;
        b1 = load(clampK(k - 4));
        LOADS_ISSUED();
        stage(k - 2, b2);
    }
    if (IPM_PRIO_SUBST != IPM_PRIO_LANE)
        SET_PRIO(IPM_PRIO_LANE);
    WAVE_SYNC();
}
#ifndef SWEEPS_VECTOR
#define SWEEPS_VECTOR 1 // substitution sweeps on v_mfma_f64_4x4x4_4b_f64, one or two columns (0: the 16-wide tile sweeps for every column count)
#endif
// the sweep for a right-hand-side specification
template <class P>
__device__ inline void factorSweepAny(const LDSP Ctx *cin, TileShared &sh, const RhsSpec &sp)
{
#if SWEEPS_VECTOR
    if (sp.n == 1)
        factorSweepFused<P, 1>(cin, sh, sp);
    else
        factorSweepFused<P, 2>(cin, sh, sp);
#else
    factorSweepFused<P, 0>(cin, sh, sp);
#endif
}
template <class P>
__device__ inline void fwdSweepAny(const LDSP Ctx *cin, const RhsSpec &sp)
{
#if SWEEPS_VECTOR
    if (sp.n == 1)
        fwdSweepV<P, 1>(cin);
    else
        fwdSweepV<P, 2>(cin);
#else
    fwdSweep<P>(cin, sp);
#endif
}
template <class P>
__device__ inline void bwdSweepAny(const LDSP Ctx *cin, const RhsSpec &sp)
{
#if SWEEPS_VECTOR
    if (sp.n == 1)
        bwdSweepV<P, 1>(cin);
    else
        bwdSweepV<P, 2>(cin);
#else
    bwdSweep<P>(cin, sp);
#endif
}

} // namespace ipm
} // namespace scpp
