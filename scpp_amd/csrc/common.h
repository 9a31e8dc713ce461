// Shared device/host definitions for the HIP kernels (gfx950 only).
// With -DSCPP_HIP_EMU the same sources build against tests/emu/hip_emu.h (CPU wave emulator used
// ONLY by the CPU-side unit tests; the product library is always built by hipcc without it).
#pragma once
#ifdef SCPP_HIP_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
typedef double d4_t __attribute__((ext_vector_type(4)));
#endif
#include <cmath>
#include <cstdint>

namespace scpp
{

constexpr int WAVE = 64;

// Every hot kernel here runs ONE 64-lane wavefront per workgroup.  Lanes of a wavefront execute in lockstep and
// the LDS / vector-memory pipelines process a wavefront's instructions in order, so exchanging data between
// lanes through LDS or global memory needs only a wavefront-scope fence (no instruction emitted) instead of
// __syncthreads() (s_waitcnt vmcnt(0) + s_barrier, which exposes every outstanding store's latency).
#ifdef SCPP_HIP_EMU
#define WAVE_SYNC() __syncthreads()
#else
#define WAVE_SYNC()                                            \
    do                                                         \
    {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)
#endif

// A pointer that is the same in every lane, re-materialised as an SGPR pair in the GLOBAL address space.
// Out-of-line device functions receive their arguments in VGPRs / through private memory, where the
// compiler must assume divergence (waterfall loops around buffer accesses) and generic addressing (flat_*).
#ifdef SCPP_HIP_EMU
template <class T>
inline T *uniformPtr(T *p) { return p; }
inline int uniformInt(int v) { return v; }
#else
template <class T>
__device__ __forceinline__ T *uniformPtr(T *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    typedef __attribute__((address_space(1))) T *gptr_t;
    return (T *)(gptr_t)((unsigned long long)(hi) << 32 | lo);
}
__device__ __forceinline__ int uniformInt(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// 1/d and 1/sqrt(d) from the hardware seed + two Newton steps (no div_scale / div_fixup special-case handling:
// the arguments are positive, normal pivots)
#ifdef SCPP_HIP_EMU
inline double fastRcp(double d) { return 1. / d; }
inline double fastRsqrt(double d) { return 1. / sqrt(d); }
#else
__device__ __forceinline__ double fastRcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ double fastRsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    double h = 0.5 * y, e = __builtin_fma(-d * y, h, 0.5);
    y = __builtin_fma(y, e, y);
    h = 0.5 * y;
    e = __builtin_fma(-d * y, h, 0.5);
    y = __builtin_fma(y, e, y);
    return y;
}
#endif

// ---- cross-lane moves on the VALU data path (DPP / v_readlane): no LDS crossbar round trip (ds_bpermute) ----
// rowXor1 / rowXor2: partner inside the quad; rowHalfMirror / rowMirror: lane 7-i of the half row / lane 15-i of the row of
// 16 lanes -- a SYMMETRIC butterfly, so that every lane of a row ends up with the bitwise identical sum (rotations would
// give each lane its own rounding, and per-lane copies of one cone scalar must agree exactly); rowRor8 = partner lane ^ 8;
// pairHead: the even lane of each pair; readLane: one lane's value for the whole wave.
#ifdef SCPP_HIP_EMU
inline double rowXor1(double v) { return __shfl_xor(v, 1); }
inline double rowXor2(double v) { return __shfl_xor(v, 2); }
inline double rowHalfMirror(double v) { const int l = threadIdx.x & 63; return __shfl(v, (l & ~7) | (7 - (l & 7))); }
inline double rowMirror(double v) { const int l = threadIdx.x & 63; return __shfl(v, (l & ~15) | (15 - (l & 15))); }
inline double rowRor8(double v) { return __shfl_xor(v, 8); }
inline double pairHead(double v) { const int l = threadIdx.x & 63; return __shfl(v, l & ~1); }
inline double readLane(double v, int src) { return __shfl(v, src); }
template <int J>
inline double rowBcast(double v) { const int l = threadIdx.x & 63; return __shfl(v, (l & ~15) | J); }
inline bool anyLane(bool p) { int v = p ? 1 : 0; for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m); return v != 0; }
inline double prevLane(double v) { const int l = threadIdx.x & 63; const double o = __shfl(v, l > 0 ? l - 1 : 0); return l > 0 ? o : 0.; }
#else
template <int CTRL>
__device__ __forceinline__ double dppMove(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int rlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int rhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double rowXor1(double v) { return dppMove<0xB1>(v); } // quad_perm:[1,0,3,2]
__device__ __forceinline__ double rowXor2(double v) { return dppMove<0x4E>(v); } // quad_perm:[2,3,0,1]
__device__ __forceinline__ double rowHalfMirror(double v) { return dppMove<0x141>(v); } // row_half_mirror
__device__ __forceinline__ double rowMirror(double v) { return dppMove<0x140>(v); }     // row_mirror
__device__ __forceinline__ double rowRor8(double v) { return dppMove<0x128>(v); } // row_ror:8
__device__ __forceinline__ double pairHead(double v) { return dppMove<0xA0>(v); } // quad_perm:[0,0,2,2]
// lane J of every row of 16 lanes to the whole row (row_newbcast, gfx90a+)
template <int J>
__device__ __forceinline__ double rowBcast(double v)
{
    return dppMove<0x150 + J>(v);
}
__device__ __forceinline__ double readLane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ bool anyLane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// value of lane - 1 (lane 0: 0): wave_shr:1 on the VALU data path (GFX9 wavefront shift)
__device__ __forceinline__ double prevLane(double v) { return dppMove<0x138>(v); }
#endif
// ---- exchanges ACROSS the four rows of 16 lanes, still on the VALU data path (gfx950: v_permlane16_swap / v_permlane32_swap) ----
// v_permlane16_swap vdst, src: rows 1 / 3 of vdst <-> rows 0 / 2 of src; v_permlane32_swap: lanes 32..63 of vdst <-> lanes 0..31 of
// src.  With both operands holding the same value v the two results are "the even rows (lower half) everywhere" and "the odd rows
// (upper half) everywhere", so two swaps broadcast any one row to all four:
//   rowGroupBcast<GS>(v): every lane (g, i) receives v of lane (GS, i);
//   rowGroupDiag(v):      every lane (g, i) receives v of lane (i & 3, i)  (the source row depends on the column: two selects).
// Each replaces an LDS write -> wait -> read -> wait round trip of the inverse-factor eliminations (tile_engine.h).
#ifdef SCPP_HIP_EMU
template <int GS>
inline double rowGroupBcast(double v) { const int l = threadIdx.x & 63; return __shfl(v, GS * 16 + (l & 15)); }
inline double rowGroupDiag(double v) { const int l = threadIdx.x & 63; return __shfl(v, (l & 3) * 16 + (l & 15)); }
#else
template <int GS>
__device__ __forceinline__ int rowGroupBcast32(int v)
{
    const auto a = __builtin_amdgcn_permlane16_swap(unsigned(v), unsigned(v), false, false); // a[0]: rows (0,0,2,2), a[1]: rows (1,1,3,3)
    const unsigned t = (GS & 1) ? a[1] : a[0];
    const auto b = __builtin_amdgcn_permlane32_swap(t, t, false, false); // b[0]: lower half twice, b[1]: upper half twice
    return int((GS & 2) ? b[1] : b[0]);
}
template <int GS>
__device__ __forceinline__ double rowGroupBcast(double v)
{
    return __hiloint2double(rowGroupBcast32<GS>(__double2hiint(v)), rowGroupBcast32<GS>(__double2loint(v)));
}
__device__ __forceinline__ int rowGroupDiag32(int v, bool odd, bool upper)
{
    const auto a = __builtin_amdgcn_permlane16_swap(unsigned(v), unsigned(v), false, false);
    const unsigned t = odd ? a[1] : a[0]; // rows (0|odd) of each half, for THIS lane's column
    const auto b = __builtin_amdgcn_permlane32_swap(t, t, false, false);
    return int(upper ? b[1] : b[0]);
}
__device__ __forceinline__ double rowGroupDiag(double v)
{
    const int i = threadIdx.x & 15;
    const bool odd = i & 1, upper = i & 2;
    return __hiloint2double(rowGroupDiag32(__double2hiint(v), odd, upper), rowGroupDiag32(__double2loint(v), odd, upper));
}
#endif
// sum / max over the row of 16 lanes, result in every lane of the row
__device__ __forceinline__ double rowSum16(double v)
{
    v += rowXor1(v);
    v += rowXor2(v);
    v += rowHalfMirror(v);
    v += rowMirror(v);
    return v;
}
__device__ __forceinline__ double rowMax16(double v)
{
    v = fmax(v, rowXor1(v));
    v = fmax(v, rowXor2(v));
    v = fmax(v, rowHalfMirror(v));
    v = fmax(v, rowMirror(v));
    return v;
}
__device__ __forceinline__ double waveMaxDpp(double v)
{
    v = rowMax16(v);
    return fmax(fmax(readLane(v, 0), readLane(v, 16)), fmax(readLane(v, 32), readLane(v, 48)));
}

// Wave reductions on the VALU data path (round 4; used by the per-iteration phases of ipm_kernel): four symmetric DPP butterflies
// inside every row of 16 lanes -- every lane of a row then holds the bitwise identical row sum -- and the four row results combined
// through v_readlane in one fixed order, (r0 + r1) + (r2 + r3).  The xor butterflies below (wave_sum / wave_max) go through the LDS
// crossbar six times per reduction (ds_bpermute: a dependent round trip each); a phase of the interior-point iteration takes up to
// eleven such reductions.  Different summation ORDER than wave_sum: equal up to rounding, same value in every lane.
__device__ __forceinline__ double waveSumDpp(double v)
{
    v = rowSum16(v);
    return (readLane(v, 0) + readLane(v, 16)) + (readLane(v, 32) + readLane(v, 48));
}
__device__ __forceinline__ int waveOrBallot(int v) { return anyLane(v != 0) ? 1 : 0; }

__device__ __forceinline__ double wave_sum(double v)
{
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int m = 32; m >= 1; m >>= 1)
    {
        const double o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_or(int v)
{
    for (int m = 32; m >= 1; m >>= 1)
        v |= __shfl_xor(v, m);
    return v;
}

// single-tangent forward dual: device-side stand-in for the reference's CppAD tape
// (scpp_core/include/systemDynamics.hpp:109-168,206-235). One lane = one seed direction.
struct Dual1
{
    double v, d;
    __host__ __device__ Dual1() : v(0.), d(0.) {}
    __host__ __device__ Dual1(double c) : v(c), d(0.) {}
    __host__ __device__ Dual1(double v_, double d_) : v(v_), d(d_) {}
};
__host__ __device__ inline Dual1 operator+(Dual1 a, Dual1 b) { return Dual1(a.v + b.v, a.d + b.d); }
__host__ __device__ inline Dual1 operator-(Dual1 a, Dual1 b) { return Dual1(a.v - b.v, a.d - b.d); }
__host__ __device__ inline Dual1 operator-(Dual1 a) { return Dual1(-a.v, -a.d); }
__host__ __device__ inline Dual1 operator*(Dual1 a, Dual1 b) { return Dual1(a.v * b.v, a.d * b.v + a.v * b.d); }
__host__ __device__ inline Dual1 operator/(Dual1 a, Dual1 b)
{
    const double inv = 1. / b.v;
    const double q = a.v * inv;
    return Dual1(q, (a.d - q * b.d) * inv);
}
__host__ __device__ inline Dual1 operator+(Dual1 a, double b) { return Dual1(a.v + b, a.d); }
__host__ __device__ inline Dual1 operator+(double a, Dual1 b) { return Dual1(a + b.v, b.d); }
__host__ __device__ inline Dual1 operator-(Dual1 a, double b) { return Dual1(a.v - b, a.d); }
__host__ __device__ inline Dual1 operator-(double a, Dual1 b) { return Dual1(a - b.v, -b.d); }
__host__ __device__ inline Dual1 operator*(Dual1 a, double b) { return Dual1(a.v * b, a.d * b); }
__host__ __device__ inline Dual1 operator*(double a, Dual1 b) { return Dual1(a * b.v, a * b.d); }
__host__ __device__ inline Dual1 operator/(Dual1 a, double b) { return Dual1(a.v / b, a.d / b); }
__host__ __device__ inline Dual1 operator/(double a, Dual1 b) { return Dual1(a) / b; }
__host__ __device__ inline Dual1 dsqrt(Dual1 a)
{
    const double r = sqrt(a.v);
    return Dual1(r, a.d * 0.5 / r);
}
__host__ __device__ inline Dual1 dsin(Dual1 a) { return Dual1(sin(a.v), a.d * cos(a.v)); }
__host__ __device__ inline Dual1 dcos(Dual1 a) { return Dual1(cos(a.v), -a.d * sin(a.v)); }
__host__ __device__ inline double dsqrt(double a) { return sqrt(a); }
__host__ __device__ inline double dsin(double a) { return sin(a); }
__host__ __device__ inline double dcos(double a) { return cos(a); }
__host__ __device__ inline double valueOf(double a) { return a; }
__host__ __device__ inline double valueOf(Dual1 a) { return a.v; }
__host__ __device__ inline double tangentOf(double) { return 0.; }
__host__ __device__ inline double tangentOf(Dual1 a) { return a.d; }

// Runge-Kutta-Fehlberg 7(8), propagated with the 8th-order weights: the scheme of the reference's
// boost::numeric::odeint::runge_kutta_fehlberg78 (discretizationImplementation.hpp:141).
constexpr int RK_S = 13;
constexpr double RK_C[RK_S] = {0., 2. / 27., 1. / 9., 1. / 6., 5. / 12., 1. / 2., 5. / 6., 1. / 6., 2. / 3., 1. / 3., 1., 0., 1.};
constexpr double RK_A[RK_S][RK_S] = {
    {0},
    {2. / 27.},
    {1. / 36., 1. / 12.},
    {1. / 24., 0., 1. / 8.},
    {5. / 12., 0., -25. / 16., 25. / 16.},
    {1. / 20., 0., 0., 1. / 4., 1. / 5.},
    {-25. / 108., 0., 0., 125. / 108., -65. / 27., 125. / 54.},
    {31. / 300., 0., 0., 0., 61. / 225., -2. / 9., 13. / 900.},
    {2., 0., 0., -53. / 6., 704. / 45., -107. / 9., 67. / 90., 3.},
    {-91. / 108., 0., 0., 23. / 108., -976. / 135., 311. / 54., -19. / 60., 17. / 6., -1. / 12.},
    {2383. / 4100., 0., 0., -341. / 164., 4496. / 1025., -301. / 82., 2133. / 4100., 45. / 82., 45. / 164., 18. / 41.},
    {3. / 205., 0., 0., 0., 0., -6. / 41., -3. / 205., -3. / 41., 3. / 41., 6. / 41., 0.},
    {-1777. / 4100., 0., 0., -341. / 164., 4496. / 1025., -289. / 82., 2193. / 4100., 51. / 82., 33. / 164., 12. / 41., 0., 1.}};
constexpr double RK_B[RK_S] = {0., 0., 0., 0., 0., 34. / 105., 9. / 35., 9. / 35., 9. / 280., 9. / 280., 0., 41. / 840., 41. / 840.};

} // namespace scpp
