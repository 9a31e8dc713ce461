// Persistent SCvx kernel of the streaming engine (round 5): ONE launch per job, one wavefront per slot, and each wavefront takes its instance
// through the whole of SCvxAlgorithm::solve (scpp_core/src/SCvxAlgorithm.cpp:61-164,166-227) by itself --
//     refill (harvest the finished instance's row, pull the next instance off the queue, cold start)
//     -> multipleShooting of its K - 1 segments, if the last candidate was accepted      (discretize_kernel.h: discretizeSegment)
//     -> sub-problem solve                                                                (ipm_solve.h: ipmSolveInstance)
//     -> nonlinear cost, accept / reject, radius update, roll-back                        (scvx_kernels.h: scvxCostUpdate)
// -- until the queue is empty.  These are the bodies of the four kernels the pool engine launches per round (scpp_hip.cpp: scvxRound),
// called in the same order on the same buffers, so every instance's arithmetic is what it was: the result rows are bitwise those of the
// pool engine and of the batch entry point (tests/test_emu_stream_fuzz.py, tests/test_gpu_parity.py).
//
// Why.  The pool engine runs rounds: every launch is a barrier for the slots of its pool, and an interior-point launch lasts as long as
// its slowest instance (5 .. 40 iterations, mean 14), so on average only ~70 % of the chip's 2048 wavefront slots held a running wavefront
// (measured).  Here a wavefront never waits for another instance: the chip holds 2048 wavefronts in different steps of
// different instances for the whole job, and the HBM-bound interior-point phases of some overlap the ALU-bound integration of others.
//
// Memory ordering inside one wavefront (there are no kernel boundaries any more): vector stores of a step are visible to the vector loads
// of the next one (same CU, write-through L1, the stores are waited for); the SCALAR data cache is not coherent with vector stores, and the
// steps read wave-uniform data the previous step wrote (the radius in ip[], the warm-start scalars, the masks) through scalar loads: it is
// invalidated at every step boundary (stepFence).
#pragma once
#include "discretize_kernel.h"
#include "ipm_solve.h"
#include "scvx_kernels.h"

namespace scpp
{

#ifndef PERSIST_COST_SPLIT
#define PERSIST_COST_SPLIT 1 // 1: scvxCostUpdateSplit (two lanes per segment) ; 0: the stand-alone kernel's one lane per segment (spills at 256 registers)
#endif
struct PersistentOut
{
    double *A, *Bm, *C, *S, *Z; // dd of the slots (the non-const view of KernelArgs' A .. Z)
    int disc_steps;
    double *shares; // [8] wavefront time (s_memtime ticks) per step, summed over wavefronts: refill, discretize, solve, cost ; may be null
};
// THE kernel parameter: one struct, so that its layout IS the kernel-argument segment.  Every step re-reads what it needs from that segment
// (constant memory, scalar loads on demand) instead of the kernel carrying ~100 pointers in SGPRs across the whole job -- the first build,
// with eight by-value parameters live across the loop, spilled 659 SGPRs into VGPR lanes, the configuration this toolchain has miscompiled
// before.  KernelArgs FIRST: ipmSolveInstance's tail reads it at offset 0 (KERNEL_TAIL_ARGS).
template <class T>
struct PersistentArgs
{
    ipm::KernelArgs a;
    SCBuffers b;
    SCvxBuffers v;
    StreamQueue q;
    typename T::Params mp;
    scpp_sc_opts sc;
    scpp_scvx_opts so;
    PersistentOut o;
};

// ---- the SC mode (SCAlgorithm::solve, scpp_core/src/SCAlgorithm.cpp:134-189) on the same pattern: a wavefront takes its instance through up to
// max_iterations rounds of multipleShooting + sub-problem solve (ipmSolveInstance with do_sc_update = 1 applies readSolution, the weight doubling and
// the convergence test, and clears `active`); no cost step, no queue ----
struct ScPersistentArgs
{
    ipm::KernelArgs a; // FIRST (ipmSolveInstance's tail)
    int B, K, max_iterations, disc_steps;
    const double *ip;            // [B][IP_N]
    double *A, *Bm, *C, *S, *Z;  // dd of the instances
    const int *active;
};
#ifdef SCPP_HIP_EMU
struct ScPersistentArgsHolder
{
    static inline const ScPersistentArgs *p = nullptr;
};
#endif

#ifdef SCPP_HIP_EMU
#define PERSIST_STEP_FN inline
template <class T>
struct PersistentArgsHolder
{
    static inline const PersistentArgs<T> *p = nullptr; // the emulator runs one kernel at a time on one host thread
};
#define PERSIST_ARGS(T, A) const PersistentArgs<T> &A = *PersistentArgsHolder<T>::p
#define SC_PERSIST_ARGS(A) const ScPersistentArgs &A = *ScPersistentArgsHolder::p
template <class S>
inline S argCopy(const S &src)
{
    return src;
}
#else
#define PERSIST_STEP_FN static __device__ __attribute__((noinline, disable_tail_calls))
// The address of the kernel-argument segment is only available in the KERNEL function (__builtin_amdgcn_kernarg_segment_ptr folds to null in a
// callee -- found as a memory fault at address 0 on the first hardware run): the kernel leaves it in LDS, the steps pick it up from there and
// make it wave-uniform again (scalar loads from constant memory).
__shared__ unsigned long long persist_kernarg_address;
template <class ARGS>
__device__ inline const __attribute__((address_space(4))) ARGS *persistentArgsLateOf()
{
    typedef const __attribute__((address_space(4))) ARGS CA;
    const unsigned long long v = persist_kernarg_address;
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    CA *p = (CA *)((unsigned long long)lo | ((unsigned long long)hi << 32));
    asm volatile("" : "+s"(p));
    return p;
}
#define PERSIST_ARGS(T, A) const __attribute__((address_space(4))) PersistentArgs<T> &A = *persistentArgsLateOf<PersistentArgs<T>>()
#define SC_PERSIST_ARGS(A) const __attribute__((address_space(4))) ScPersistentArgs &A = *persistentArgsLateOf<ScPersistentArgs>()
// a member struct of the kernel-argument segment, copied word by word (scalar loads from constant memory)
template <class S>
__device__ __forceinline__ S argCopy(const __attribute__((address_space(4))) S &src)
{
    static_assert(sizeof(S) % 4 == 0, "argument structs are multiples of a dword");
    S dst;
    const __attribute__((address_space(4))) unsigned *ps = (const __attribute__((address_space(4))) unsigned *)&src;
    unsigned *pd = reinterpret_cast<unsigned *>(&dst);
#pragma unroll
    for (unsigned i = 0; i < sizeof(S) / 4; i++)
        pd[i] = ps[i];
    return dst;
}
#endif

__device__ __forceinline__ void stepFence()
{
#ifndef SCPP_HIP_EMU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the step's stores have left the wavefront
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_dcache_inv();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
    WAVE_SYNC();
}
__device__ __forceinline__ int uniformLoad(const int *p)
{
    return uniformInt(*(const volatile int *)p);
}

// ---- the four steps, each a function of its own (own register allocation, nothing live across a step but the slot index) ----
template <class T>
PERSIST_STEP_FN void persistRefill(long slot)
{
    PERSIST_ARGS(T, A);
    const SCBuffers b = argCopy(A.b);
    const SCvxBuffers v = argCopy(A.v);
    const StreamQueue q = argCopy(A.q);
    const typename T::Params mp = argCopy(A.mp);
    const scpp_sc_opts sc = argCopy(A.sc);
    const scpp_scvx_opts so = argCopy(A.so);
    scvxStreamRefill<T>(b, v, q, mp, sc, so, slot);
    stepFence();
}
template <class T, class Model, bool FOH>
PERSIST_STEP_FN void persistDiscretize(long slot)
{
    PERSIST_ARGS(T, A);
    const int B = A.b.B, K = A.b.K, steps = A.o.disc_steps;
    const double *X = A.a.X, *U = A.a.U, *sigma = A.a.sigma, *par = A.b.ip + ipm::IP_PAR;
    double *Ao = A.o.A, *Bo = A.o.Bm, *Co = A.o.C, *So = A.o.S, *Zo = A.o.Z;
    // the integration's LDS is the dynamic region the solver's LDS-resident segment fields use during a solve (never live together)
#ifdef SCPP_HIP_EMU
    static DiscLds<Model, FOH, false> lds_obj;
    DiscLds<Model, FOH, false> *lds = &lds_obj;
#else
    extern __shared__ __attribute__((aligned(16))) double seg_lds[]; // 16-byte base: DiscLds members are aligned(16) (b128 LDS accesses), whatever the static LDS before it adds up to
    DiscLds<Model, FOH, false> *lds = reinterpret_cast<DiscLds<Model, FOH, false> *>(seg_lds);
#endif
    for (int k = 0; k < K - 1; k++)
    {
        discretizeSegment<Model, FOH, false>(B, K, X, U, sigma, par, ipm::IP_N, nullptr, Ao, Bo, Co, So, Zo, steps, slot, k, lds);
        WAVE_SYNC(); // the segment's last LDS reads before the next segment's first writes
    }
    stepFence();
}
template <class T, class P>
PERSIST_STEP_FN void persistSolve(long slot)
{
    PERSIST_ARGS(T, A);
    const ipm::KernelArgs a = argCopy(A.a);
    ipm::ipmSolveInstance<P>(a, int(slot), &A.a);
    if (IPM_PRIO_LANE)
        SET_PRIO(0); // the integration and the cost step run at the default priority
    stepFence();
}
template <class T, class Model>
PERSIST_STEP_FN void persistCost(long slot)
{
    PERSIST_ARGS(T, A);
    const SCBuffers b = argCopy(A.b);
    const SCvxBuffers v = argCopy(A.v);
    const scpp_scvx_opts so = argCopy(A.so);
    // two lanes per segment (half the stage slopes per lane: they fit this kernel's 256 registers); the per-segment sums pass through 64 doubles of
    // the dynamic LDS region, which is idle during this step
#if PERSIST_COST_SPLIT
#ifdef SCPP_HIP_EMU
    static double seg_sum[WAVE];
#else
    extern __shared__ __attribute__((aligned(16))) double seg_lds[]; // 16-byte base: DiscLds members are aligned(16) (b128 LDS accesses), whatever the static LDS before it adds up to
    double *seg_sum = seg_lds;
#endif
    scvxCostUpdateSplit<Model>(b, v, so, slot, seg_sum);
#else
    scvxCostUpdate<Model>(b, v, so, slot);
#endif
    stepFence();
}

// T: refill traits (scvx_kernels.h), Model: the flow-map plugin, P: the solver's table
template <class T, class Model, class P, bool FOH>
__global__ void __launch_bounds__(WAVE, IPM_WAVES_PER_SIMD) __attribute__((disable_tail_calls)) scvx_persistent_kernel(PersistentArgs<T> args)
{
#ifdef SCPP_HIP_EMU
    PersistentArgsHolder<T>::p = &args;
#else
    if (threadIdx.x == 0)
        persist_kernarg_address = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    WAVE_SYNC();
#endif
    const long slot = blockIdx.x;
    if (slot >= args.b.B)
        return;
    long long t_ref = 0, t_disc = 0, t_ipm = 0, t_cost = 0;
    for (;;)
    {
        PERSIST_ARGS(T, A);
        const long long t0 = clock64();
        persistRefill<T>(slot); // acts on an empty / terminated slot only
        if (uniformLoad(A.b.active + slot) == 0)
            break; // the queue is drained and nothing is running in this slot
        const long long t1 = clock64();
        if (uniformLoad(A.v.needs_disc + slot) != 0)
            persistDiscretize<T, Model, FOH>(slot);
        const long long t2 = clock64();
        persistSolve<T, P>(slot);
        const long long t3 = clock64();
        persistCost<T, Model>(slot);
        const long long t4 = clock64();
        t_ref += t1 - t0;
        t_disc += t2 - t1;
        t_ipm += t3 - t2;
        t_cost += t4 - t3;
    }
#ifndef SCPP_HIP_EMU
    {
        PERSIST_ARGS(T, A);
        if (A.o.shares && threadIdx.x == 0)
        {
            unsigned long long *sh = reinterpret_cast<unsigned long long *>(A.o.shares);
            atomicAdd(sh + 0, (unsigned long long)t_ref);
            atomicAdd(sh + 1, (unsigned long long)t_disc);
            atomicAdd(sh + 2, (unsigned long long)t_ipm);
            atomicAdd(sh + 3, (unsigned long long)t_cost);
        }
    }
#endif
}

// ---- SC mode ----
template <class Model, bool FOH, bool VT>
PERSIST_STEP_FN void scPersistDiscretize(long slot)
{
    SC_PERSIST_ARGS(A);
    const int B = A.B, K = A.K, steps = A.disc_steps;
    const double *X = A.a.X, *U = A.a.U, *sigma = A.a.sigma, *par = A.ip + ipm::IP_PAR;
    double *Ao = A.A, *Bo = A.Bm, *Co = A.C, *So = A.S, *Zo = A.Z;
#ifdef SCPP_HIP_EMU
    static DiscLds<Model, FOH, VT> lds_obj;
    DiscLds<Model, FOH, VT> *lds = &lds_obj;
#else
    extern __shared__ __attribute__((aligned(16))) double seg_lds[]; // 16-byte base: DiscLds members are aligned(16) (b128 LDS accesses), whatever the static LDS before it adds up to
    DiscLds<Model, FOH, VT> *lds = reinterpret_cast<DiscLds<Model, FOH, VT> *>(seg_lds);
#endif
    for (int k = 0; k < K - 1; k++)
    {
        discretizeSegment<Model, FOH, VT>(B, K, X, U, sigma, par, ipm::IP_N, nullptr, Ao, Bo, Co, So, Zo, steps, slot, k, lds);
        WAVE_SYNC();
    }
    stepFence();
}
template <class P>
PERSIST_STEP_FN void scPersistSolve(long slot)
{
    SC_PERSIST_ARGS(A);
    const ipm::KernelArgs a = argCopy(A.a);
    ipm::ipmSolveInstance<P>(a, int(slot), &A.a);
    stepFence();
}
template <class Model, class P, bool FOH, bool VT>
__global__ void __launch_bounds__(WAVE, IPM_WAVES_PER_SIMD) __attribute__((disable_tail_calls)) sc_persistent_kernel(ScPersistentArgs args)
{
#ifdef SCPP_HIP_EMU
    ScPersistentArgsHolder::p = &args;
#else
    if (threadIdx.x == 0)
        persist_kernarg_address = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    WAVE_SYNC();
#endif
    const long slot = blockIdx.x;
    if (slot >= args.B)
        return;
    const int max_iterations = args.max_iterations;
    for (int it = 0; it < max_iterations; it++)
    {
        SC_PERSIST_ARGS(A);
        if (uniformLoad(A.active + slot) == 0)
            break; // converged, failed, or at its iteration limit (the solve's tail clears the flag)
        scPersistDiscretize<Model, FOH, VT>(slot);
        scPersistSolve<P>(slot);
    }
}

} // namespace scpp
