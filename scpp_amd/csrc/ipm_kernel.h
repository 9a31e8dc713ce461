// Batched structure-exploiting interior-point solver for the SC / SCvx sub-problem of any model described by a
// constraint table (constraint_table.h).  Replaces, for B problem instances at once, the reference's per-iteration
//   solver->solve(false)      scpp_core/src/SCAlgorithm.cpp:78   (Epigraph -> ECOS)
// on the problem of buildSCProblem (scpp_core/src/SCProblem.cpp:6-138) + Model::addApplicationConstraints
// (scpp_models/src/rocketQuat.cpp:70-144, rocket2d.cpp:46-84), followed by readSolution + the convergence/weight logic of
// SCAlgorithm::iterate (SCAlgorithm.cpp:100-131).
//
// Formulation (scalar twin with the derivation: oracle/structured_ipm.hpp):
//   * presolve: the variables the table pins (x_0, fixed final-state components, fixed final inputs, states / inputs pinned
//     for the whole horizon) are constants; <= 16 stage variables w_k = (x_k[XMAP], u_k[UMAP]) remain, plus delta_k, nu_k,
//     nu_bound_k, sigma, delta_sigma, norm1_nu;
//   * primal-dual Mehrotra predictor-corrector with Nesterov-Todd scaling (the ECOS scheme), no
//     self-dual embedding (virtual control makes every sub-problem feasible);
//   * per IPM iteration ONE factorisation of the reduced KKT system: nu, nu_bound, norm1_nu, delta_k,
//     delta_sigma eliminated in closed form, then a block-tridiagonal quasi-definite system
//       [H_k M_k'; M_k -E_k^-1] ... coupled by N_k, with sigma as a one-column border,
//     factorised stage by stage with dense 16x16 Cholesky tiles (FP64 MFMA v_mfma_f64_16x16x4_f64 for the tile products).
//
// Mapping: ONE 64-lane wavefront (one workgroup) per problem instance. Element-wise phases run with
// lane == stage/segment index (K <= 64); the factorisation / substitution sweeps are sequential over
// stages with the 64 lanes cooperating on the 16x16 tiles.  All per-instance state lives in an HBM
// workspace (layout below).
#pragma once
#include "common.h"
#include "cone_math.h"
#include "constraint_table.h"
#include <type_traits>
#include <utility>

namespace scpp
{
namespace ipm
{

#ifndef IPM_FAC_ALIGN
#define IPM_FAC_ALIGN 1 // 0: the factor / saved-column blocks where the field-major records end and records of 472 doubles (until round 6; A/B hook)
#endif
// ---- record layout of a model's problem (doubles per stage / segment) ----
template <class P>
struct Lay : Derived<P>
{
    using D = Derived<P>;
    static constexpr int NX = P::NX, NU = P::NU, NXV = P::NXV, NUV = P::NUV, NS = D::NS, NL = D::NL, HS_N = D::HS_N;
    // ---- stage record ----
    static constexpr int F_W = 0;                 // [16] stage variables
    static constexpr int F_DL = F_W + NV;         // delta_k
    static constexpr int F_DW = F_DL + 1;         // [16]
    static constexpr int F_DDL = F_DW + NV;
    static constexpr int F_WBAR = F_DDL + 1;      // [16] trust-region centre
    static constexpr int F_UHAT = F_WBAR + NV;    // [3]
    static constexpr int F_HDD = F_UHAT + 3;
    static constexpr int F_HDW = F_HDD + 1;       // [16]
    static constexpr int F_RXW = F_HDW + NV;      // [16]
    static constexpr int F_RXD = F_RXW + NV;
    static constexpr int F_S = F_RXD + 1;         // [NS]
    static constexpr int F_Z = F_S + NS;
    static constexpr int F_DS = F_Z + NS;
    static constexpr int F_DZ = F_DS + NS;
    static constexpr int F_RZ = F_DZ + NS;
    static constexpr int F_TZ = F_RZ + NS;
    static constexpr int F_LS = F_TZ + NS;        // lambda (scaled)
    static constexpr int F_DSS = F_LS + NS;       // (W^-1 ds_aff) o (W dz_aff): conic product of the scaled affine directions
    static constexpr int F_ETA = F_DSS + NS;      // [NCONES]
    static constexpr int F_WB = F_ETA + D::NCONES; // [LP0] wbar of the cones, same offsets as the slack layout
    static constexpr int F_BXW = F_WB + D::LP0;   // [16] right-hand side (w part)
    static constexpr int F_BXD = F_BXW + NV;
    static constexpr int F_WBK = F_BXD + 1;       // [17] W and delta of the last iterate that met the reduced tolerances
    static constexpr int STREC = F_WBK + NV + 1;
    // ---- exchange record, STAGE-major: sx[k][XREC].  Everything the tile sweeps read or write per stage as single entries
    //      (right-hand sides in, solutions out, the Hessian inputs of the factorisation).  In the field-major records each of
    //      these ~210 accesses per stage and iteration touched its own 128-byte line for 8 useful bytes; here a stage's
    //      entries are 10 consecutive lines, and the lane = stage phases reach them with one stride-XREC access per field.
    static constexpr int X_BETA = 0;              // [16] condensed right-hand side (w part)         phases -> sweeps
    static constexpr int X_BCW = X_BETA + NV;     // [16] border column of the block solve (w part)  sweeps -> phases
    static constexpr int X_VW = X_BCW + NV;       // [16] block-solve output (w part)                sweeps -> phases
    static constexpr int X_HS = X_VW + NV;        // [HS_N] small Hessian blocks of the application cones / LP rows
    static constexpr int X_HC = X_HS + HS_N;      // [2]  {1/eta1^2, 2/(2 w0^2 - 1)} of the trust-region cone
    static constexpr int X_WBT = X_VW + NV + 32;  // [16] wbar_1..16 of the trust-region cone (copy of F_WB + 1 ..)
    static constexpr int X_RHO = X_WBT + NV;      // [NL] condensed right-hand side (multiplier part)
    static constexpr int X_BCL = X_RHO + 16;      // [NL] border column (multiplier part)
    static constexpr int X_VL = X_BCL + 16;       // [NL] block-solve output (multiplier part)
    static constexpr int X_EINV = X_VL + 16;      // [NL] E^-1 of the segment
    static constexpr int X_S = X_EINV + 16;       // [NL] S_k of the segment (copy of dd: right-hand side of the border column)
    static constexpr int XREC = X_S + 16;         // 176 doubles = eleven 128-byte lines
    static_assert(HS_N + 2 <= 32 && NL <= 16 && XREC % 16 == 0, "exchange record layout");
    // ---- segment record: G_NFIELDS fields of NL doubles (enum SegField) ----
    // ---- per-stage factor record, PACKED (the kernel is HBM-throughput bound): Li lower triangle (136), Yt 16 x NL,
    //      Ti lower triangle of the NL x NL block.  Z = Ti N is not stored: N = [I | -C] makes it 4 extra matrix-core
    //      instructions from Ti and the entries of C.
    static constexpr int FAC_LI = 0, FAC_YT = NV * (NV + 1) / 2, FAC_TI = FAC_YT + NV * NL;
    // Multiple of a 128-byte line, and the block starts on one (facOffset below): a stage's record then spans exactly FACREC / 16 lines for each of its
    // four passes per iteration (one write, three reads).  Until round 6 the block started wherever the field-major records ended (16 bytes past a
    // line at K = 50) and a record of 472 doubles = 29.5 lines: every record straddled one line more than it fills, and so did every 128-byte vector
    // of the saved forward columns behind it -- 51 KB per iteration in lines for 25.6 KB of data (tools/traffic_table.py, the emulator's tracer).
    static constexpr int FACREC = IPM_FAC_ALIGN ? ((FAC_TI + NL * (NL + 1) / 2 + 15) & ~15) : ((FAC_TI + NL * (NL + 1) / 2 + 7) & ~7);
    // field-major copy of the segment dynamics (A, B, C, s, z) for the lane = segment phases
    static constexpr int DY_A = 0, DY_B = NX * NX, DY_C = DY_B + NX * NU, DY_S = DY_C + NX * NU, DY_Z = DY_S + NX, DYNREC = DY_Z + NX;
};
// ---- segment record fields (each NL doubles) ----
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
    G_QV,
    G_BTN,
    G_BNB,
    G_DINV,
    G_CV,
    G_BXNU,
    G_BXNUB,
    G_BY,
    G_NFIELDS
};
constexpr int NRHS_MAX = 3;            // right-hand-side columns carried by one sweep
constexpr int SVREC = 2 * NRHS_MAX * 16; // saved forward intermediates (a, c) per stage

// Stage / segment records are stored FIELD-major: element f of stage k lives at st[f*pitch + k], so that the
// lane == stage phases read and write fully coalesced rows (one lane per stage).
constexpr int LANES = 64;
// row pitch (doubles) of the field-major records: one row = one field of all K stages.  K rounded up to even
// instead of 64 keeps rows 16-byte aligned and cuts the record traffic by 1 - K/64 (22 % at K = 50).
__host__ __device__ inline int recPitch(int K) { return (K + 1) & ~1; }
constexpr int GSAVE = 16; // wave-uniform scalars of the last solve (sigma, delta_sigma, n1 and their slacks / duals): warm start
constexpr int RSAVE = 112; // resume block of the split schedule (ipm_split.h): Glob + Iter + the loop's locals between two launches
// extra doubles between the workspaces of consecutive instances (a multiple of 16: measurement hook for the address-interleaving experiments of round 6)
#ifndef IPM_WS_PAD
#define IPM_WS_PAD 0
#endif
static_assert(IPM_WS_PAD % 16 == 0, "workspaces start on a 128-byte line");
// doubles from the start of an instance's workspace to its factor records: exchange records, then the three field-major blocks, rounded up to a line
template <class P>
__host__ __device__ inline size_t facOffset(int K)
{
    using L = Lay<P>;
    const size_t n = size_t(K) * L::XREC + size_t(recPitch(K)) * (L::STREC + G_NFIELDS * L::NL + L::DYNREC);
    return IPM_FAC_ALIGN ? ((n + 15) & ~size_t(15)) : n;
}
template <class P>
__host__ __device__ inline size_t workspaceDoubles(int K)
{
    using L = Lay<P>;
    // the exchange records come first so that they start on a 128-byte line; the total is a multiple of a line
    const size_t n = facOffset<P>(K) + size_t(K) * (L::FACREC + SVREC) + GSAVE + RSAVE;
    return ((n + 15) & ~size_t(15)) + IPM_WS_PAD;
}

// Strided view of one lane's record, addressed through a buffer resource: every access is
//   buffer_load/store_dwordx2 v, v_lane_byte_offset, s[rsrc:rsrc+3], s_field_offset offen
// i.e. ONE 32-bit VGPR (lane*8) serves all fields and the field offsets live in SGPRs.  (Plain global
// pointers made the compiler keep ~70 loop-invariant 64-bit VGPR addresses -- one per 4 KB window of the
// field-major record -- and spill them.)
// byte offset beyond every record block of an instance: a buffer load at it returns 0, a buffer store is dropped
constexpr int VO_OOB = 0x40000000;
typedef unsigned int u32x2_t __attribute__((vector_size(8)));
// cache policy of the workspace accesses (aux operand of the buffer instructions; gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef IPM_LD_AUX
#define IPM_LD_AUX 0
#endif
#ifndef IPM_ST_AUX
#define IPM_ST_AUX 0
#endif
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
            return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lb, so, IPM_LD_AUX));
        }
        __device__ const Ref &operator=(double x) const
        {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, x), rsrc, lb, so, IPM_ST_AUX);
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
// the same view type over the stage-major exchange records: lane offset = stage * XREC, field pitch 8 bytes
__device__ inline SV makeSX(double *block, int xrec, int K, unsigned stage)
{
    return SV{__builtin_amdgcn_make_buffer_rsrc(block, 0, K * xrec * 8, 0x00020000), int(stage * unsigned(xrec) * 8u), 0, 8};
}

// End of a load group: nothing is scheduled across this point, so every load written above it is issued before the arithmetic
// below starts (left alone, the pressure heuristics of the scheduler sink loads into the arithmetic and turn one memory round
// trip per group into one per handful of values -- counted in the ISA with tools/isa_round_trips.py).
// issue priority of a wavefront inside a lane phase (s_setprio; the sweeps set their own, sweeps.h): measurement hook of round 6
#ifndef IPM_PRIO_LANE
#define IPM_PRIO_LANE 0
#endif
#ifndef SCPP_HIP_EMU // (the emulator's traffic tracer, tests/emu/hip_emu.h: which phase a buffer access belongs to, which record block it hits)
#define EMU_PHASE(name)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        if (IPM_PRIO_LANE)                                                                                             \
            __builtin_amdgcn_s_setprio(IPM_PRIO_LANE);                                                                 \
    } while (0)
#define EMU_TRAFFIC_REGION(name, base, bytes, field_bytes, rec_bytes)
#define EMU_TRAFFIC_MANUAL(what, n, store)
#endif
#ifdef SCPP_HIP_EMU
#define LOADS_ISSUED()
#else
#define LOADS_ISSUED() __builtin_amdgcn_sched_barrier(0)
#endif

struct Settings
{
    double feastol, abstol, reltol, gamma;
    int maxit;
    int use_mfma;
};

// LDS address space of the wave-uniform state and of the LDS-resident segment fields (plain pointers in the CPU emulation)
#ifdef SCPP_HIP_EMU
#define LDSP
#else
#define LDSP __attribute__((address_space(3)))
#endif
// Segment fields that live in LDS for the duration of the main loop (round 4).  gfx950 has 160 KB of LDS per CU = 20 KB for each of
// the 8 wavefronts the register file admits, of which the kernel used 3.6 KB; the workspace on the other hand streams from HBM in
// every phase.  The three most-travelled fields of the segment records -- lam (read 8 x, written once per
// interior-point iteration), nu (7 + 1) and nu_b (6 + 1): 24 of the 71 field accesses per row -- are copied into LDS after the
// initialisation, used from there by every per-iteration phase, and copied back when the solve ends (the warm start of the next
// launch reads them from the workspace).  Element (slot, row i, segment k) lives at segl[(slot * NL + i) * pitch + k]: a lane = segment
// access is one conflict-free ds_read_b64 / ds_write_b64.  K = 50: 3 x 14 x 50 x 8 = 16 800 B + 3.0 KB static <= 20 480 B.
enum SegLdsSlot
{
    SL_LAM = 0,
    SL_NU,
    SL_NUB,
    NSEGLDS
};
// everything a wavefront needs to know about its instance
struct Ctx
{
    LDSP double *segl; // [NSEGLDS][NL][pitch] LDS-resident segment fields (main loop only)
    int K, lane;
    int pitch;   // row pitch of the field-major records (doubles)
    double *st;  // [STREC][pitch]   field-major
    double *sg;  // [SEGREC][pitch]  field-major
    double *dy;  // [DYNREC][pitch]  field-major copy of A,B,C,s,z
    double *fac; // [K][FACREC]
    double *sv;  // [K][SVREC]
    double *sx;  // [K][XREC]        stage-major exchange records (sweeps <-> phases)
    double *gsave; // [GSAVE]
    const double *A, *B, *C, *S, *Z; // dd of this instance
    const double *ip;                // instance parameters
};

// dynamic LDS of ipm_kernel<P> (bytes): the LDS-resident segment fields
template <class P>
__host__ __device__ inline unsigned segLdsBytes(int K)
{
    return SegInLds<P>::value ? unsigned(NSEGLDS * Lay<P>::NL * recPitch(K) * 8) : 0u;
}

// ---------------- the table, evaluated ----------------
// Every index into the table is a template constant (sfor hands the loop counter over as an integral_constant), so that the
// front end folds the table away completely: what reaches the optimiser is the same straight-line code a hand-written
// saff / Lmul / LTmul for the model would be.  (Plain unrolled loops over the constexpr arrays left the folding to late
// optimisation passes, after the per-lane slack arrays had already been demoted to scratch memory.)
template <class F, int... Is>
__device__ inline void sforImpl(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ inline void sfor(F &&f)
{
    sforImpl(f, std::make_integer_sequence<int, N>{});
}
#define SFOR_IDX(name, tag) constexpr int name = decltype(tag)::value

// coefficient of a term at this node
template <int COEF, class AU>
__device__ inline double termCoefC(const double *ip, AU uh)
{
    if constexpr (COEF == CF_ONE)
        return 1.;
    else if constexpr (COEF == CF_UHAT0)
        return uh[0];
    else if constexpr (COEF == CF_UHAT1)
        return uh[1];
    else if constexpr (COEF == CF_UHAT2)
        return uh[2];
    else
        return ip[COEF];
}
// row(w) = sum_t mul_t coef_t w[var_t]   (+ hmul * ip[hpar] if WITH_CONST), summed in table order, constant last
template <class P, int C, int I, bool WITH_CONST, class AW, class AU>
__device__ inline double rowValue(const double *ip, AW w, AU uh)
{
    // C >= 0: row I of application cone C ; C == -1: LP row I
    constexpr Row r = C >= 0 ? P::CONES[C >= 0 ? C : 0].r[I < MAXDIM ? I : 0] : P::LPS[I < P::NLP ? I : 0];
    double s = 0.;
    if constexpr (r.t[0].var >= 0)
    {
        constexpr Term t0 = r.t[0];
        s = (t0.coef == CF_ONE && t0.mul == 1.) ? double(w[t0.var]) : (t0.mul == 1. ? termCoefC<t0.coef>(ip, uh) * w[t0.var] : t0.mul * termCoefC<t0.coef>(ip, uh) * w[t0.var]);
        if constexpr (r.t[1].var >= 0)
        {
            constexpr Term t1 = r.t[1];
            s += (t1.coef == CF_ONE && t1.mul == 1.) ? double(w[t1.var]) : (t1.mul == 1. ? termCoefC<t1.coef>(ip, uh) * w[t1.var] : t1.mul * termCoefC<t1.coef>(ip, uh) * w[t1.var]);
        }
        if constexpr (r.t[2].var >= 0)
        {
            constexpr Term t2 = r.t[2];
            s += (t2.coef == CF_ONE && t2.mul == 1.) ? double(w[t2.var]) : (t2.mul == 1. ? termCoefC<t2.coef>(ip, uh) * w[t2.var] : t2.mul * termCoefC<t2.coef>(ip, uh) * w[t2.var]);
        }
    }
    if constexpr (WITH_CONST && r.hpar >= 0)
    {
        const double h = r.hmul == 1. ? ip[r.hpar] : (r.hmul == -1. ? -ip[r.hpar] : r.hmul * ip[r.hpar]);
        if constexpr (r.t[0].var >= 0)
            s += h;
        else
            s = h;
    }
    return s;
}

template <class P, class AV>
__device__ inline void maskInactive(unsigned act, AV v)
{
    using D = Derived<P>;
    sfor<P::NCONE>([&](auto ct) {
        SFOR_IDX(C, ct);
        constexpr int OFF = D::coneOff(C + 1), DIM = P::CONES[C].dim;
        if (!(act & (1u << (C + 1)))) // (the trust-region cone, bit 0, is active at every node)
            sfor<DIM>([&](auto it) { v[OFF + decltype(it)::value] = 0.; });
    });
    sfor<P::NLP>([&](auto lt) {
        SFOR_IDX(Lr, lt);
        if (!(act & (1u << (D::NCONES + Lr))))
            v[D::LP0 + Lr] = 0.;
    });
}
// affine slack h - Gx of one stage
template <class P, class AW, class AB, class AU, class AO>
__device__ inline void saff(const double *ip, unsigned act, AW wk, double dlk, AB wb, AU uh, AO out)
{
    using D = Derived<P>;
    const bool scvx = ip[IP_SCVX] != 0.;
    out[0] = scvx ? ip[IP_TR] : dlk;
    sfor<NV>([&](auto jt) {
        SFOR_IDX(j, jt);
        if constexpr (j >= D::NVU)
            out[1 + j] = 0.;
        else if constexpr (j < P::NXV)
            out[1 + j] = scvx ? 0. : wb[j] - wk[j];
        else
            out[1 + j] = wb[j] - wk[j];
    });
    sfor<P::NCONE>([&](auto ct) {
        SFOR_IDX(C, ct);
        constexpr int OFF = D::coneOff(C + 1);
        sfor<P::CONES[C].dim>([&](auto it) {
            SFOR_IDX(I, it);
            out[OFF + I] = rowValue<P, C, I, true>(ip, wk, uh);
        });
    });
    sfor<P::NLP>([&](auto lt) {
        SFOR_IDX(Lr, lt);
        out[D::LP0 + Lr] = rowValue<P, -1, Lr, true>(ip, wk, uh);
    });
    maskInactive<P>(act, out);
}
// linear part of saff
template <class P, class AW, class AU, class AO>
__device__ inline void Lmul(const double *ip, unsigned act, AW dwk, double ddlk, AU uh, AO out)
{
    using D = Derived<P>;
    const bool scvx = ip[IP_SCVX] != 0.;
    out[0] = scvx ? 0. : ddlk;
    sfor<NV>([&](auto jt) {
        SFOR_IDX(j, jt);
        if constexpr (j >= D::NVU)
            out[1 + j] = 0.;
        else if constexpr (j < P::NXV)
            out[1 + j] = scvx ? 0. : -dwk[j];
        else
            out[1 + j] = -dwk[j];
    });
    sfor<P::NCONE>([&](auto ct) {
        SFOR_IDX(C, ct);
        constexpr int OFF = D::coneOff(C + 1);
        sfor<P::CONES[C].dim>([&](auto it) {
            SFOR_IDX(I, it);
            out[OFF + I] = rowValue<P, C, I, false>(ip, dwk, uh);
        });
    });
    sfor<P::NLP>([&](auto lt) {
        SFOR_IDX(Lr, lt);
        out[D::LP0 + Lr] = rowValue<P, -1, Lr, false>(ip, dwk, uh);
    });
    maskInactive<P>(act, out);
}
// contribution of term T of row (C, I) to (L'v)[J]: accumulates into add in table order
template <class P, int J, int C, int I, int T, class AV, class AU>
__device__ inline void ltTerm(const double *ip, AV v, AU uh, double &add, bool &any)
{
    using D = Derived<P>;
    constexpr Row r = C >= 0 ? P::CONES[C >= 0 ? C : 0].r[I < MAXDIM ? I : 0] : P::LPS[I < P::NLP ? I : 0];
    constexpr Term tm = r.t[T];
    if constexpr (tm.var == J)
    {
        constexpr int SL = C >= 0 ? D::coneOff((C >= 0 ? C : 0) + 1) + I : D::LP0 + I;
        const double vi = v[SL];
        const double x = (tm.coef == CF_ONE && tm.mul == 1.) ? vi : (tm.mul == 1. ? termCoefC<tm.coef>(ip, uh) * vi : tm.mul * termCoefC<tm.coef>(ip, uh) * vi);
        add = any ? add + x : x;
        any = true;
    }
}
// L' v (entries of inactive cones must be zero)
template <class P, class AV, class AU>
__device__ inline void LTmul(const double *ip, unsigned fm, AV v, AU uh, double *gw, double *gdl)
{
    using D = Derived<P>;
    const bool scvx = ip[IP_SCVX] != 0.;
    *gdl = v[0];
    sfor<NV>([&](auto jt) {
        SFOR_IDX(J, jt);
        double add = 0.;
        bool any = false;
        sfor<P::NCONE>([&](auto ct) {
            SFOR_IDX(C, ct);
            sfor<P::CONES[C].dim>([&](auto it) {
                SFOR_IDX(I, it);
                ltTerm<P, J, C, I, 0>(ip, v, uh, add, any);
                ltTerm<P, J, C, I, 1>(ip, v, uh, add, any);
                ltTerm<P, J, C, I, 2>(ip, v, uh, add, any);
            });
        });
        sfor<P::NLP>([&](auto lt) {
            SFOR_IDX(Lr, lt);
            ltTerm<P, J, -1, Lr, 0>(ip, v, uh, add, any);
            ltTerm<P, J, -1, Lr, 1>(ip, v, uh, add, any);
            ltTerm<P, J, -1, Lr, 2>(ip, v, uh, add, any);
        });
        double g;
        if constexpr (J >= D::NVU)
            g = 0.;
        else if constexpr (J < P::NXV)
            g = scvx ? 0. : -v[1 + J];
        else
            g = -v[1 + J];
        if (any)
            g += add;
        gw[J] = (fm & (1u << J)) ? 0. : g;
    });
}

// dynamics residual of segment k on the field-major copy (coalesced, fully unrolled: all loads of a row are in flight
// together):  x_{k+1} - A x_k - B u_k - C u_{k+1} - S sigma - nu_k - Z_k     (pinned states / inputs are 0)
template <class P, class A0, class A1, class AN>
__device__ inline void dynResF(const SV &dy, A0 w0, A1 w1, AN nuv, double sig, double *out)
{
    using L = Lay<P>;
    double x0[NV], u1[P::NUV];
#pragma unroll
    for (int j = 0; j < NV; j++)
        x0[j] = w0[j];
#pragma unroll
    for (int j = 0; j < P::NUV; j++)
        u1[j] = w1[P::NXV + j];
    sfor<P::NX>([&](auto rowt) {
        SFOR_IDX(i, rowt);
        constexpr int xi = L::XINV.v[i];
        double acc = (xi >= 0 ? double(w1[xi >= 0 ? xi : 0]) : 0.) - dy[L::DY_S + i] * sig - nuv[i] - dy[L::DY_Z + i];
        sfor<P::NXV>([&](auto jt) {
            SFOR_IDX(j, jt);
            acc -= dy[L::DY_A + i * P::NX + P::XMAP[j]] * x0[j];
        });
        sfor<P::NUV>([&](auto jt) {
            SFOR_IDX(j, jt);
            acc -= dy[L::DY_B + i * P::NU + P::UMAP[j]] * x0[P::NXV + j] + dy[L::DY_C + i * P::NU + P::UMAP[j]] * u1[j];
        });
        out[i] = acc;
    });
}

// Per-stage Hessian data (delta_k eliminated) for the in-sweep tile build: F_HDD, F_HDW (delta elimination),
// X_HC = {1/eta1^2, 2/den} of the trust-region cone and X_HS = the small dense blocks contributed by the application
// cones / LP rows (layout: Derived<P>::PAT).
template <class P>
__device__ inline int hsIndex(int a, int b)
{
    return Derived<P>::PAT.idx[a][b];
}
// Hs += sum_ab c_a c_b W^-2_ab e_va e_vb'   (rows of a cone are single terms or constants)
template <class P, int C, class AW>
__device__ inline void addConeHs(const double *ip, double *Hs, double eta, AW w)
{
    using D = Derived<P>;
    constexpr int d = P::CONES[C].dim;
    const double e2 = 1. / (eta * eta);
    sfor<d>([&](auto at) {
        SFOR_IDX(a, at);
        constexpr Term ta = P::CONES[C].r[a].t[0];
        if constexpr (ta.var >= 0)
        {
            const double va = (a == 0) ? double(w[0]) : -double(w[a]);
            const double ca = ta.coef == CF_ONE ? ta.mul : ta.mul * ip[ta.coef >= 0 ? ta.coef : 0];
            sfor<d>([&](auto bt) {
                SFOR_IDX(b, bt);
                constexpr Term tb = P::CONES[C].r[b].t[0];
                if constexpr (tb.var >= 0)
                {
                    const double vb = (b == 0) ? double(w[0]) : -double(w[b]);
                    const double cb = tb.coef == CF_ONE ? tb.mul : tb.mul * ip[tb.coef >= 0 ? tb.coef : 0];
                    double Wab = 2. * va * vb;
                    if constexpr (a == b)
                        Wab += (a == 0) ? -1. : 1.;
                    constexpr int idx = D::PAT.idx[ta.var][tb.var];
                    Hs[idx] += ca * cb * Wab * e2;
                }
            });
        }
    });
}
template <class P>
__device__ inline void buildHs(const Ctx &c, int k, bool identity)
{
    using L = Lay<P>;
    const unsigned fm = L::fixedMask(k, c.K), act = L::activeMask(k, c.K);
    const SV st = makeSV(c.st, L::STREC, unsigned(k), c.pitch);
    const SV eta = st + L::F_ETA, wb = st + L::F_WB, uh = st + L::F_UHAT;
    const SV xs = makeSX(c.sx, L::XREC, c.K, unsigned(k));
    double Hs[L::HS_N > 0 ? L::HS_N : 1];
    for (int i = 0; i < L::HS_N; i++)
        Hs[i] = 0.;
    {
        const bool scvx = c.ip[IP_SCVX] != 0.;
        // load group first: a store between two loads pins their order (one memory round trip per entry otherwise)
        // (SCvx: the state rows 1 .. NXV of wbar are structural zeros and are read through an out-of-range view, ipm_solve.h: padView)
        const SV stz = SV{st.rsrc, scvx ? VO_OOB : st.lb, st.fo, st.pb};
        const SV wbz = stz + L::F_WB;
        double w[NV + 1];
#pragma unroll
        for (int j = 0; j <= NV; j++)
            w[j] = (j >= 1 && j <= P::NXV) ? double(wbz[j]) : double(wb[j]);
        const double e0 = eta[0];
        const double e2 = 1. / (e0 * e0);
        const double den = 2. * w[0] * w[0] - 1.;
        // SC: delta_k eliminated -> H = e2 (I - (2/den) w w').  SCvx: delta_k constant -> plain L'W^-2 L = e2 (I + 2 w w')
        // on the rows that exist (the mask is applied where the tile is built, sweeps.h buildHTile)
        st[L::F_HDD] = scvx ? 1. : den * e2;
#pragma unroll
        for (int j = 0; j < NV; j++)
        {
            stz[L::F_HDW + j] = ((fm & (1u << j)) || scvx) ? 0. : 2. * w[0] * w[1 + j] * e2; // SCvx: stays 0 (cleared by phSetup)
            xs[L::X_WBT + j] = w[1 + j];
        }
        xs[L::X_HC] = e2;
        xs[L::X_HC + 1] = scvx ? -2. : 2. / den;
    }
    sfor<P::NCONE>([&](auto ct) {
        SFOR_IDX(C, ct);
        constexpr int OFF = L::coneOff(C + 1);
        if (act & (1u << (C + 1)))
            addConeHs<P, C>(c.ip, Hs, eta[C + 1], wb + OFF);
    });
    sfor<P::NLP>([&](auto lt) {
        SFOR_IDX(Lr, lt);
        if (act & (1u << (L::NCONES + Lr)))
        {
            const double d = identity ? 1. : st[L::F_Z + L::LP0 + Lr] / st[L::F_S + L::LP0 + Lr];
            sfor<3>([&](auto at) {
                SFOR_IDX(ta, at);
                constexpr Term a = P::LPS[Lr].t[ta];
                if constexpr (a.var >= 0)
                    sfor<3>([&](auto bt) {
                        SFOR_IDX(tb, bt);
                        constexpr Term b = P::LPS[Lr].t[tb];
                        if constexpr (b.var >= 0)
                        {
                            constexpr int idx = L::PAT.idx[a.var][b.var];
                            constexpr bool unit = a.coef == CF_ONE && b.coef == CF_ONE && a.mul * b.mul == 1.;
                            if constexpr (unit)
                                Hs[idx] += d;
                            else
                                Hs[idx] += d * (a.mul == 1. ? termCoefC<a.coef>(c.ip, uh) : a.mul * termCoefC<a.coef>(c.ip, uh)) *
                                           (b.mul == 1. ? termCoefC<b.coef>(c.ip, uh) : b.mul * termCoefC<b.coef>(c.ip, uh));
                        }
                    });
            });
        }
    });
    for (int i = 0; i < L::HS_N; i++)
        xs[L::X_HS + i] = Hs[i];
}

} // namespace ipm
} // namespace scpp
