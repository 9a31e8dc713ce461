// Register-resident 16x16 FP64 tile engine for the block-tridiagonal KKT sweeps of ipm_kernel.
//
// A tile lives in the MFMA C/D layout of v_mfma_f64_16x16x4_f64 ("D-layout"): lane l = (g = l>>4, i = l&15)
// holds T[g + 4r][i] in register r = 0..3.  With that layout the product X'Y of two tiles is FOUR matrix-core
// instructions straight from registers (operand r of both tiles feeds MFMA r; contraction index g + 4r), and
// the result is again a D-layout tile -- so every stage operation of the factorisation and of the
// substitution sweeps is expressed as X'Y:
//   Yt = Li M' = mm(Lit, Mt)     Theta = E^-1 + Yt'Yt = mm(Yt, Yt)     Z = Ti N = mm(Tit, N)     Phi' = H' + mm(Z, Z)
//   forward :  a = mm(Lit, g)   gl = rho - mm(Yt, a)   c = mm(Tit, gl)   g' = beta' + mm(Z, c)       (multi-RHS columns)
//   backward:  t = mm(Zt, x') - c   lam = mm(Ti, t)   s = a - mm(Y, lam)   x = mm(Li, s)
// Only the two inverse Cholesky factors per stage (Li = chol(Phi)^-1, Ti = chol(Theta)^-1) are not MFMA work:
// they are computed by Gaussian elimination on [A | I] (rank-1 updates of both tiles, all 64 lanes busy, the
// pivot row fetched by ds_bpermute, the pivot column by DPP row broadcast).  Scalar twin: oracle/structured_ipm.hpp.
#pragma once
#include "ipm_kernel.h"

namespace scpp
{
namespace ipm
{

struct Tile
{
    double v[4];
};

__device__ inline Tile tileZero()
{
    Tile t;
    t.v[0] = t.v[1] = t.v[2] = t.v[3] = 0.;
    return t;
}

// X'Y on the FP64 matrix core (4 x v_mfma_f64_16x16x4_f64), optionally accumulating into C
__device__ inline Tile mm(const Tile &X, const Tile &Y)
{
    d4_t acc = {0., 0., 0., 0.};
#pragma unroll
    for (int r = 0; r < 4; r++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X.v[r], Y.v[r], acc, 0, 0, 0);
    Tile t;
    t.v[0] = acc[0];
    t.v[1] = acc[1];
    t.v[2] = acc[2];
    t.v[3] = acc[3];
    return t;
}

// row-major 16x16 global tile <-> D-layout registers
__device__ inline Tile loadTile(const double *p, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
        t.v[r] = p[(g + 4 * r) * 16 + i];
    return t;
}
__device__ inline Tile loadTileT(const double *p, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Tile t;
#pragma unroll
    for (int r = 0; r < 4; r++)
        t.v[r] = p[i * 16 + (g + 4 * r)];
    return t;
}
__device__ inline void storeTile(double *p, int lane, const Tile &t)
{
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; r++)
        p[(g + 4 * r) * 16 + i] = t.v[r];
}

// ---- matrix x VECTOR on the small-block FP64 matrix instruction (round 4) ----
// The substitution sweeps of a single right-hand-side column multiply 16 x 16 tiles with ONE vector, five times per stage and in a
// dependent chain.  As X'Y on v_mfma_f64_16x16x4_f64 that is four 16-pass instructions (64 cycles each, measured) per product for one
// useful column of sixteen.  v_mfma_f64_4x4x4_4b_f64 computes four independent 4 x 4 x 4 blocks in 4 passes, and a 16 x 16 matrix-vector
// product is exactly four such instructions: block b of instruction t multiplies T[4b .. 4b+3][4t .. 4t+3] with x[4t .. 4t+3], the
// accumulator carries the sum over t.  Register layouts (measured on gfx950, tests/tools/mfma4x4_probe.hip: A lane = i + 4b + 16k,
// B lane = j + 4b + 16k, D lane = j + 4b + 16i):
//   V layout of a 16-vector:  lane l holds x[e(l)],  e(l) = 4 ((l >> 2) & 3) + (l >> 4)       (each element in four lanes, l & 3 free);
//   A4 layout of a matrix T:  register t of lane l holds T[l & 15][4t + (l >> 4)] -- which IS the D layout (tile layout of the 16-wide
//                             instruction: register r of lane (g, i) holds X[g + 4r][i]) of X = T', so  mv(X, x) = X' x  for every tile X;
//   B operand of instruction t = x[4t + (l >> 4)] = the V-layout value of lane 4t of every row of 16 lanes: one DPP row broadcast.
// The result is again in V layout, so products chain without any re-arrangement.
#ifdef SCPP_HIP_EMU
inline double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
#else
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
#endif
__device__ inline int vElem(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
// y = X' x   (X: a D-layout tile, i.e. X' in A4 layout; x and y in V layout) -- the vector counterpart of mm(X, Y) = X'Y
__device__ inline double mv(const Tile &T, double x)
{
    double acc = mfma4(T.v[0], rowBcast<0>(x), 0.);
    acc = mfma4(T.v[1], rowBcast<4>(x), acc);
    acc = mfma4(T.v[2], rowBcast<8>(x), acc);
    acc = mfma4(T.v[3], rowBcast<12>(x), acc);
    return acc;
}

// ---- packed factor storage ----
__device__ inline int triIdx(int row, int col) { return (row * (row + 1)) / 2 + col; }
#ifndef INVCHOL_BPERMUTE
#define INVCHOL_BPERMUTE 1
#endif
#ifndef INVCHOL_PERMLANE
#define INVCHOL_PERMLANE 0
#endif
struct TileShared
{
#if !INVCHOL_BPERMUTE && !INVCHOL_PERMLANE
    double colA[2][16]; // pivot column / row exchange of the LDS variant of the eliminations only (640 B the LDS-resident
    double rowR[2][16]; // segment fields need: 8 wavefronts x 20 KB fill the CU's 160 KB exactly)
    double pv[16];
#endif
    double tr[16 * 17]; // transpose scratch
#ifdef IPM_PROFILE
    double prof[10]; // factor sweep: cycles in the two eliminations, whole sweep, calls, stage head, between the eliminations; the stage head split:
                     // tail of the previous stage (after the second elimination), issuing the loads, H tile, Z'Z
#endif
};

// transpose of a D-layout tile ON THE MATRIX CORE (round 4): mm(X, I) = X' I, exact in floating point (every product is x * 1 or x * 0),
// four instructions of a pipe the kernel leaves mostly idle, no LDS traffic and no wavefront fences on the stage's dependent chain
__device__ inline Tile transposeTileMfma(const Tile &t, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Tile id;
#pragma unroll
    for (int r = 0; r < 4; r++)
        id.v[r] = (g + 4 * r == i) ? 1. : 0.;
    return mm(t, id);
}
#ifndef TRANSPOSE_MFMA
#define TRANSPOSE_MFMA 0 // measured (round 4, same box, bitwise identical): 5063 / 5149 / 5121 against 5091 / 5135 / 5136 converged/s with the LDS
                         // transpose -- no difference; the LDS path stays the default
#endif
// transpose a D-layout tile through LDS
__device__ inline Tile transposeTile(const Tile &t, TileShared &sh, int lane)
{
#if TRANSPOSE_MFMA
    (void)sh;
    return transposeTileMfma(t, lane);
#endif
    const int g = lane >> 4, i = lane & 15;
    WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 4; r++)
        sh.tr[(g + 4 * r) * 17 + i] = t.v[r];
    WAVE_SYNC();
    Tile o;
#pragma unroll
    for (int r = 0; r < 4; r++)
        o.v[r] = sh.tr[i * 17 + (g + 4 * r)];
    return o;
}

// Li = chol(A)^-1 (lower triangular, D-layout) of the SPD leading n x n block of tile A; rows/cols >= n: identity.
// Gaussian elimination on [A | I], fully unrolled so that everything that depends only on the step index is
// resolved at compile time (which register holds pivot row j, which register rows lie entirely above /
// below the pivot).  Per step: the pivot row of A and of R by ds_bpermute from the row group that holds it (INVCHOL_BPERMUTE;
// the older exchange through 2 x 16 doubles of LDS remains selectable), the pivot column by DPP row broadcast, a Newton
// reciprocal, <= 8 predicated FMAs.
// Inlined into its callers: as a called function its entry waits for every outstanding memory operation of the wavefront
// (s_waitcnt vmcnt(0) is part of the function-call ABI), which exposes the latency of the stage loads the factor sweep issues
// to run under the elimination, and of the factor-record stores in front of the second elimination.
#ifndef INVCHOL_LINKAGE
#define INVCHOL_LINKAGE __device__ inline __attribute__((always_inline))
#endif
#ifndef INVCHOL_UNROLL
#define INVCHOL_UNROLL _Pragma("unroll")
#endif
#ifndef INVCHOL_PERMLANE
#define INVCHOL_PERMLANE 0 // 0: pivot row / column exchanged through 2 x 16 doubles of LDS ; 1: by v_permlane16/32_swap (no LDS at all;
                           // measured on MI355X, round 3: 3930 vs 4046 converged/s -- 8 swaps + their hazard s_nops per step issue more than the
                           // LDS round trip costs, the eliminations are not LDS-latency bound)
#endif
#ifndef INVCHOL_BPERMUTE
#define INVCHOL_BPERMUTE 1 // 1 (default since round 3): the pivot ROW of A and of R is fetched from the row group that holds it by
                           // ds_bpermute -- ONE trip through the LDS crossbar, nothing stored -- instead of publish -> wait -> read (two
                           // trips).  The kernel is latency-bound (SQ counters): 4420 / 4423 against 4313 / 4367
                           // converged/s on the same box (+1.9 %).  Uses A[j][i] where the LDS exchange (-DINVCHOL_BPERMUTE=0) uses
                           // A[i][j]: equal up to rounding, and the scalar twin's own formula (structured_ipm.hpp: invCholFactor)
#endif
#ifndef INVCHOL_PIVOTS_AT_END
#define INVCHOL_PIVOTS_AT_END 1
#endif
#ifndef INVCHOL_SPECULATE
#define INVCHOL_SPECULATE 1
#endif
// FLOOR = false: the elimination WITHOUT the pivot floor (no floor read, no v_max between a pivot and its reciprocal: five instructions
// less per step, and the eliminations are issue-bound at two wavefronts per SIMD); `*ok` then says whether every pivot was above its
// floor, in which case the result is bitwise what the floored elimination returns.  invCholFactor runs this one first and repeats with
// FLOOR = true in the (rare) other case.
template <int n, bool FLOOR>
INVCHOL_LINKAGE Tile invCholImpl(Tile A, TileShared &sh, int lane, bool *ok)
{
    const int g = lane >> 4, i = lane & 15;
    Tile R;
    double od = 0.;     // original diagonal entry of MY column (for the pivot floor)
    double pvr[4];      // pivots of my rows
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        R.v[r] = (g + 4 * r == i) ? 1. : 0.;
        pvr[r] = 1.;
        od = (g + 4 * r == i) ? A.v[r] : od;
    }
    // pivot floor 1e-14 A[i][i] of column i: held by lane (i & 3, i), read by v_readlane at step i (the eliminations of the
    // 8 wavefronts of a CU are bound by the bandwidth of the one LDS they share: everything wave-uniform stays out of it)
    od *= 1e-14;
    // compile-time step index: which register holds pivot row j, which registers lie entirely above the pivot, and the DPP
    // control word of the row broadcast are all constants of the step
    sfor<n>([&](auto jt) {
        constexpr int j = decltype(jt)::value;
        constexpr int b = j & 1, rj_ = j >> 2;
        // The pivot column is published with ZEROS in rows <= j: the pivot-row entries A[j][i] = A[i][j] the readers take from
        // it (and the multipliers m = A[row][j] / d, which come by row broadcast) are then structurally zero wherever the
        // elimination must not act, and no reader has to mask them (rows >= n of an n < 16 block are zero in column j anyway;
        // R[j][i] = 0 for i > j by construction).
#if !INVCHOL_PERMLANE && !INVCHOL_BPERMUTE
        if (i == j)
        {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * r + 3 >= j) // rows above the pivot are never read
                {
                    const int row = g + 4 * r;
                    sh.colA[b][row] = row > j ? A.v[r] : 0.;
                }
        }
        if (g == (j & 3))
            sh.rowR[b][i] = rj_ == 0 ? R.v[0] : rj_ == 1 ? R.v[1] : rj_ == 2 ? R.v[2] : R.v[3];
        WAVE_SYNC();
#endif
        // one batch of LDS reads (no control flow in between)
        // (taking the maximum in the pivot's own lane before ONE broadcast -- two readlanes and two canonicalising v_max less per
        //  step, bitwise the same pivot -- was measured in round 3: 3915 vs 3923 converged/s, no difference; not kept)
        double d = readLane(rj_ == 0 ? A.v[0] : rj_ == 1 ? A.v[1] : rj_ == 2 ? A.v[2] : A.v[3], (j & 3) * 16 + j); // A[j][j]
        double floor_ = 0.;
        if (FLOOR)
            floor_ = readLane(od, (j & 3) * 16 + j);
#if INVCHOL_BPERMUTE
        // lane (j & 3, i) holds A[j][i] and R[j][i] in register j >> 2: every lane of column i fetches them from there
        (void)b;
        (void)sh;
        const int src = (j & 3) * 16 + i;
        // NOT masked to the columns right of the pivot: columns <= j of A are dead from step j on (no later step reads them, Li is
        // built from R and the pivots), so updating them with finite garbage is harmless -- and without the select the compiler
        // no longer waits for the crossbar before it starts the reciprocal chain (measured: the 14-step elimination ran 390 cycles
        // per step against 206 of the 16-step one, whose schedule happened to overlap the two)
        const double aj = __shfl(rj_ == 0 ? A.v[0] : rj_ == 1 ? A.v[1] : rj_ == 2 ? A.v[2] : A.v[3], src);
        const double rj = __shfl(rj_ == 0 ? R.v[0] : rj_ == 1 ? R.v[1] : rj_ == 2 ? R.v[2] : R.v[3], src); // R[j][i] (0 for i > j)
#elif !INVCHOL_PERMLANE
        const double aj = sh.colA[b][i];  // A[j][i] for i > j, 0 otherwise
        const double rj = sh.rowR[b][i];  // R[j][i] (0 for i > j)
#endif
        // column j below the pivot, for this lane's rows: lane (g, j) holds it -> row broadcast on the VALU data path
        double cr[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            cr[r] = 0.;
            if (4 * r + 3 > j)
                cr[r] = rowBcast<j>((g + 4 * r > j) ? A.v[r] : 0.);
        }
#if INVCHOL_PERMLANE
        // No LDS in the elimination at all (gfx950 row swaps, common.h).  A[i][j] (= the pivot-row entry of column i, 0 for
        // i <= j) is one of the column-j values the row broadcast just delivered: lane (i & 3, i) holds it as cr[i >> 2], and
        // rowGroupDiag hands every lane of column i that lane's value.  R[j][i] lives in row group j & 3, register j >> 2.
        // Bitwise the same operands as the LDS exchange (which remains available: -DINVCHOL_PERMLANE=0).
        (void)b;
        const int iq = i >> 2;
        const double aj = rowGroupDiag(iq == 0 ? cr[0] : iq == 1 ? cr[1] : iq == 2 ? cr[2] : cr[3]);
        const double rj = rowGroupBcast<(j & 3)>(rj_ == 0 ? R.v[0] : rj_ == 1 ? R.v[1] : rj_ == 2 ? R.v[2] : R.v[3]);
#endif
        // (round 4, measured and not kept: the floor taken off the dependent chain -- reciprocal of the raw pivot straight from the
        //  v_readlane, 1 / floor precomputed per column, a select afterwards; bitwise identical, two v_max_f64 less on the chain but
        //  seven instructions more per step: 4845 against 5128 converged/s on the same box (-5.5 %).  At two wavefronts per SIMD the
        //  eliminations are bound by instruction ISSUE: every instruction added to a step costs, every one removed pays)
        if (FLOOR)
            d = fmax(d, floor_);
        const double p = fastRcp(d);
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            if (4 * r + 3 > j) // some row of this register lies below the pivot
            {
                const double m = cr[r] * p; // 0 for rows <= j
                A.v[r] -= m * aj;
                R.v[r] -= m * rj;
            }
#if !INVCHOL_PIVOTS_AT_END
            if (r == rj_)
                pvr[r] = (g + 4 * r == j) ? d : pvr[r];
#endif
        }
    });
#if INVCHOL_PIVOTS_AT_END
    // The pivots are read off the final A (round 4): row j of A is not touched after step j - 1 (its own multipliers are 0), so the
    // diagonal lane (row & 3, row) still holds the raw pivot of its row, and its `od` is that row's floor: one v_max there, then every
    // lane of a row fetches its row's pivot from the diagonal lane -- instead of two v_cndmask in every one of the n steps.  Bitwise
    // the same value as max(d, floor) taken at the step.
    {
        const int q = i >> 2;
        const double dg = q == 0 ? A.v[0] : q == 1 ? A.v[1] : q == 2 ? A.v[2] : A.v[3]; // A[i][i] where (i & 3) == g
        const double dm = FLOOR ? fmax(dg, od) : dg;
        if (!FLOOR) // every pivot strictly above its floor (NaN fails): otherwise the caller repeats with the floor
            *ok = !anyLane((i & 3) == g && i < n && !(dg > od));
#pragma unroll
        for (int r = 0; r < 4; r++)
            pvr[r] = (4 * r < n) ? __shfl(dm, g * 16 + g + 4 * r) : 1.;
    }
#endif
    Tile Li;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int row = g + 4 * r;
        if (row < n)
            Li.v[r] = (i <= row) ? R.v[r] * fastRsqrt(pvr[r]) : 0.;
        else
            Li.v[r] = (i == row) ? 1. : 0.;
    }
    return Li;
}
template <int n>
INVCHOL_LINKAGE Tile invCholFactor(Tile A, TileShared &sh, int lane)
{
#if INVCHOL_SPECULATE && INVCHOL_PIVOTS_AT_END
    bool ok = true;
    Tile Li = invCholImpl<n, false>(A, sh, lane, &ok);
    if (!ok) // a pivot at or below 1e-14 of its original diagonal entry (or not finite): the floored elimination, as before
        Li = invCholImpl<n, true>(A, sh, lane, &ok);
    return Li;
#else
    bool ok = true;
    return invCholImpl<n, true>(A, sh, lane, &ok);
#endif
}

// The TRANSPOSED inverse factor Lit = (chol(A)^-1)' directly (round 4; selectable, NOT the default -- see INVCHOL_TRANSPOSED below).  The factor sweep needs Li only as the first operand of
// X'Y products -- mm(Lit, .) = Li . -- so until round 3 every inverse factor went through a transpose in LDS (4 writes, 4 reads, two
// waits) on the stage's dependent chain: two of them per stage = 1.0 k of the 7.6 k cycles of that chain.  Here the
// elimination keeps R' instead of R: lane (g, i) holds Rt[c][i] = R[i][c], c = g + 4r, and step j is
//      Rt[c][i] -= m_i Rt[c][j],   m_i = A[j][i] p  (i > j; the pivot-row entry the A update fetches anyway),   Rt[c][j] by DPP row broadcast,
// i.e. the ds_bpermute pair that fetched R[j][i] is replaced by row broadcasts on the VALU data path as well: half the LDS-crossbar
// traffic of an elimination, and no LDS transpose at all.  (m_i uses A[j][i] where the R update of invCholFactor uses A[i][j]: equal up
// to rounding.)
// Memory layout of the stored factor is unchanged: the caller stores Lit through the transposed offsets (sweeps.h: offTriT).
template <int n>
INVCHOL_LINKAGE Tile invCholFactorT(Tile A, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    Tile Rt;
    double od = 0.;  // original diagonal entry of MY column (for the pivot floor)
    double pvc = 1.; // pivot of row i (my column of Lit)
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        Rt.v[r] = (g + 4 * r == i) ? 1. : 0.;
        od = (g + 4 * r == i) ? A.v[r] : od;
    }
    od *= 1e-14;
    sfor<n>([&](auto jt) {
        constexpr int j = decltype(jt)::value;
        constexpr int rj_ = j >> 2;
        double d = readLane(rj_ == 0 ? A.v[0] : rj_ == 1 ? A.v[1] : rj_ == 2 ? A.v[2] : A.v[3], (j & 3) * 16 + j); // A[j][j]
        const double floor_ = readLane(od, (j & 3) * 16 + j);
        // lane (j & 3, i) holds A[j][i] in register j >> 2: every lane of column i fetches it from there (not masked, see invCholFactor)
        const double aj = __shfl(rj_ == 0 ? A.v[0] : rj_ == 1 ? A.v[1] : rj_ == 2 ? A.v[2] : A.v[3], (j & 3) * 16 + i);
        double cr[4], rc[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            cr[r] = rc[r] = 0.;
            if (4 * r + 3 > j) // column j below the pivot, for this lane's rows
                cr[r] = rowBcast<j>((g + 4 * r > j) ? A.v[r] : 0.);
            if (4 * r <= j) // R[j][c] for this lane's c = g + 4r (0 for c > j, 1 for c = j: by construction)
                rc[r] = rowBcast<j>(Rt.v[r]);
        }
        d = fmax(d, floor_);
        const double p = fastRcp(d);
        const double mi = (i > j) ? aj * p : 0.; // multiplier of row i
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            if (4 * r + 3 > j)
                A.v[r] -= (cr[r] * p) * aj;
            if (4 * r <= j)
                Rt.v[r] -= mi * rc[r];
        }
        pvc = (i == j) ? d : pvc;
    });
    const double sc = fastRsqrt(pvc);
    Tile Lit;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int c = g + 4 * r;
        if (i < n)
            Lit.v[r] = (c <= i) ? Rt.v[r] * sc : 0.;
        else
            Lit.v[r] = (c == i) ? 1. : 0.;
    }
    return Lit;
}
#ifndef INVCHOL_TRANSPOSED
#define INVCHOL_TRANSPOSED 0 // 1: the factor sweep takes Lit / Tit straight from invCholFactorT (no LDS transpose, half the ds_bpermute).  MEASURED, NOT KEPT
                              // (round 4, same box, three alternating runs): 4551 / 4563 / 4564 against 4585 / 4619 / 4606 converged/s (-1 %).  With two
                              // wavefronts per SIMD an elimination step is ISSUE-bound (42 instructions x 4 cycles x 2 waves = the 340 cycles measured):
                              // the eight extra DPP moves per step cost more issue slots than the two LDS transposes per stage cost latency
#endif

#define INVCHOL invCholFactor

} // namespace ipm
} // namespace scpp
