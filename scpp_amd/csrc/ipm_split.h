// The interior-point solve of ipm_solve.h split BY REGISTER BUDGET into two kernels per interior-point iteration (round 5).
//
// Why.  ipm_kernel<P> runs a whole solve in one resident wavefront; its register allocation is the factor sweep's (~250 VGPRs of
// tiles), which admits two wavefronts per SIMD -- while the lane phases and the vector substitution sweeps (60 % of an iteration, all of
// it memory round trips at that occupancy) need a third of those registers.  One kernel cannot have two occupancies.  Here the SAME
// phase functions are driven by two kernels:
//   ipm_split_kernel<W>   everything except the factor sweep, __launch_bounds__(64, IPM_SPLIT_WAVES = 3): from the point after a
//                         factor sweep up to the next factor sweep request (the tail of iteration i and the head of iteration i + 1 in
//                         one launch), W = SegFieldsInWorkspace<P> (12 wavefronts per CU leave no room for LDS-resident fields);
//   ipm_factor_kernel<P>  factorSweepAny<P> of every instance that asked for one, two wavefronts per SIMD as before.
// The host enqueues  A(first) F A F A ... A  for a solve (scpp_hip.cpp: launchIpmSplit); an instance that has finished returns at the
// top of every later launch.  Between two launches an instance's wave-uniform state (Glob 352 B, Iter 328 B, the loop's locals) lives
// in the RESUME block of its workspace (ipm_kernel.h: RSAVE); everything else was in the workspace already.
//
// What must stay identical: the arithmetic.  Both drivers call the same phases in the same order on the same records, so the split
// solve is BITWISE equal to ipm_kernel<P>'s (tests/test_emu_split.py on the emulator, tests/test_gpu_parity.py on hardware); the
// control flow below is ipm_kernel's main loop (ipm_solve.h) written as a resumable state machine, statement for statement.
#pragma once
#include "ipm_solve.h"

namespace scpp
{
namespace ipm
{

#ifndef IPM_SPLIT_WAVES
#define IPM_SPLIT_WAVES 3
#endif

// resume points (RS_PC of the resume block)
enum SplitPc
{
    PC_START = 0,  // set-up of an attempt (warm or cold)
    PC_INIT_F = 1, // waiting for the factor sweep of the cold initialisation
    PC_MAIN_F = 2, // waiting for the factor sweep of a main-loop iteration
    PC_NORMS = 3,  // (internal) after the initialisation
    PC_HEAD = 4,   // (internal) top of a main-loop iteration
    PC_END = 5,    // (internal) an attempt has ended
    PC_DONE = 6    // the solve has ended, outputs written
};
// layout of the resume block (doubles): Glob, Iter, then the loop's wave-uniform locals
constexpr int RS_GLOB = 0, RS_ITER = RS_GLOB + int(sizeof(Glob) / 8), RS_LOC = RS_ITER + int(sizeof(Iter) / 8);
enum SplitLocal
{
    RL_PC = 0,
    RL_SPEC,
    RL_WARM,
    RL_STATUS,
    RL_ITER,
    RL_ITER_TOTAL,
    RL_USE_BACKUP,
    RL_INACC_OK,
    RL_BK_PREV,
    RL_PRES_PREV,
    RL_N
};
static_assert(sizeof(Glob) % 8 == 0 && sizeof(Iter) % 8 == 0 && RS_LOC + RL_N <= RSAVE, "resume block layout");

template <class P>
__device__ inline Ctx splitCtx(const KernelArgs &a, int inst, int lane)
{
    using L = Lay<P>;
    constexpr int NX = P::NX, NU = P::NU;
    const int K = a.K;
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
    c.segl = nullptr;
    return c;
}

// wave-uniform struct <-> resume block, 8 bytes per lane (one coalesced access)
template <class T>
__device__ inline void saveUniform(double *dst, const PRIV T *src, int lane)
{
    const PRIV double *s = (const PRIV double *)src;
    for (int i = lane; i < int(sizeof(T) / 8); i += WAVE)
        dst[i] = s[i];
}
template <class T>
__device__ inline void loadUniform(PRIV T *dst, const double *src, int lane)
{
    PRIV double *d = (PRIV double *)dst;
    for (int i = lane; i < int(sizeof(T) / 8); i += WAVE)
        d[i] = src[i];
}

// ---- finalising pass of a TRUNCATED schedule (round 6) ----
// With fewer (factor, rest) launch pairs than the worst case 2 maxit + 1 -- scpp_hip_set_ipm_schedule(ctx, SCPP_IPM_SPLIT, pairs) or SCPP_IPM_SPLIT_PAIRS,
// the measurement hooks -- an instance can still be waiting for a factor sweep (PC_INIT_F / PC_MAIN_F) when the last launch ends.  It has written no
// outputs: status, X / U and the warm flag would keep the previous solve's values and, in SC mode, sc_iters and active would not move -- stale results
// reported as successes.  This pass retires such an instance as an iteration-limit failure (status -1, no warm start, SC loop stopped), which is what
// the resident kernel reports when it runs out of iterations.  One lane per instance; nothing to do for an instance that finished.
template <class P>
__global__ void __launch_bounds__(WAVE) ipm_split_finalize_kernel(KernelArgs a)
{
    const int inst = blockIdx.x * WAVE + threadIdx.x;
    if (inst >= a.B)
        return;
    if (a.active && a.active[inst] == 0)
        return;
    double *rl = splitCtx<P>(a, inst, 0).gsave + GSAVE + RS_LOC; // the loop's locals in the resume block
    const int pc = int(rl[RL_PC]);
    if (pc != PC_INIT_F && pc != PC_MAIN_F)
        return;
    rl[RL_PC] = PC_DONE;
    a.status[inst] = -1;
    if (a.warm)
        a.warm[inst] = 0;
    if (a.do_sc_update)
    {
        a.sc_iters[inst] += 1;
        a.active[inst] = 0;
    }
}

// ---- kernel A: everything but the factor sweep ----
template <class W>
__global__ void __launch_bounds__(WAVE, IPM_SPLIT_WAVES) __attribute__((disable_tail_calls)) ipm_split_kernel(KernelArgs a, int first)
{
    using L = Lay<W>;
    constexpr int NX = W::NX, NU = W::NU;
    static_assert(!SegInLds<W>::value, "the split kernel keeps every segment field in the workspace");
    const int inst = blockIdx.x;
    if (inst >= a.B)
        return;
    if (a.active && a.active[inst] == 0)
        return;
    const int K = a.K, lane = threadIdx.x, k = lane;
    const Ctx c = splitCtx<W>(a, inst, lane);
    double *rs = c.gsave + GSAVE; // resume block
    double *rl = rs + RS_LOC;
    int pc = PC_START;
    if (!first)
    {
        pc = uniformInt(int(rl[RL_PC]));
        if (pc != PC_INIT_F && pc != PC_MAIN_F)
            return; // finished (or never started)
    }
    const Settings opt = a.opt;

    __shared__ Glob g;
    __shared__ Iter it;
    __shared__ Ctx cshared;
    PRIV Glob *gp = (PRIV Glob *)&g;
    PRIV Iter *itp = (PRIV Iter *)&it;
    const PRIV Ctx *cs = (const PRIV Ctx *)&cshared;
    if (lane == 0)
        cshared = c;
    int warm = 0, status = -1, iter = 0, iter_total = 0, spec_n = 1;
    bool use_backup = false, inacc_ok = false, bk_prev = false;
    double pres_prev = 0.;
    if (first)
    {
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
        it.common_step = 0; // (the split SCHEDULE -- a measurement mode -- does not repeat a failed attempt with the common step length)
        it.bk_valid = 0;
        it.bk_sig = it.bk_dsg = it.bk_n1 = it.pres_prev = 0.;
        if (a.Xold && k < K)
        {
            const size_t o = size_t(inst) * K + k;
            for (int j = 0; j < NX; j++)
                a.Xold[o * NX + j] = a.X[o * NX + j];
            for (int j = 0; j < NU; j++)
                a.Uold[o * NU + j] = a.U[o * NU + j];
        }
        warm = (a.warm && a.warm[inst] != 0) ? 1 : 0;
    }
    else
    {
        loadUniform(gp, rs + RS_GLOB, lane);
        loadUniform(itp, rs + RS_ITER, lane);
        spec_n = uniformInt(int(rl[RL_SPEC]));
        warm = uniformInt(int(rl[RL_WARM]));
        status = uniformInt(int(rl[RL_STATUS]));
        iter = uniformInt(int(rl[RL_ITER]));
        iter_total = uniformInt(int(rl[RL_ITER_TOTAL]));
        use_backup = uniformInt(int(rl[RL_USE_BACKUP])) != 0;
        inacc_ok = uniformInt(int(rl[RL_INACC_OK])) != 0;
        bk_prev = uniformInt(int(rl[RL_BK_PREV])) != 0;
        pres_prev = rl[RL_PRES_PREV];
        WAVE_SYNC();
    }
    // hand over to ipm_factor_kernel: the wave-uniform state goes to the resume block
    auto yield = [&](int next_pc, int n) {
        WAVE_SYNC();
        saveUniform(rs + RS_GLOB, gp, lane);
        saveUniform(rs + RS_ITER, itp, lane);
        if (lane == 0)
        {
            rl[RL_PC] = double(next_pc);
            rl[RL_SPEC] = double(n);
            rl[RL_WARM] = double(warm);
            rl[RL_STATUS] = double(status);
            rl[RL_ITER] = double(iter);
            rl[RL_ITER_TOTAL] = double(iter_total);
            rl[RL_USE_BACKUP] = use_backup ? 1. : 0.;
            rl[RL_INACC_OK] = inacc_ok ? 1. : 0.;
            rl[RL_BK_PREV] = bk_prev ? 1. : 0.;
            rl[RL_PRES_PREV] = pres_prev;
        }
    };

    for (;;)
    {
        if (pc == PC_START)
        {
            // (ipm_kernel: top of the attempt loop)
            // (ipm_solve.h: a warm-started solve of SCAlgorithm's sub-problem takes the common step length; this schedule -- a measurement mode -- does not
            // repeat a failed cold attempt with it.  `it` lives in the resume block between launches, the flag with it)
            it.common_step = (IPM_SPLIT_STEPS && warm && c.ip[IP_SCVX] == 0.) ? 1 : 0;
            phSetup<W>(cs, a.X + size_t(inst) * K * NX, a.U + size_t(inst) * K * NU, a.uhat + size_t(inst) * K * 3, gp, itp, warm, 0);
            if (warm)
            {
                phWarmInit<W>(cs, gp, itp);
                pc = PC_NORMS;
            }
            else
            {
                phInitPrimalRhs<W>(cs, gp, itp);
                yield(PC_INIT_F, 2); // factorSweepAny(specBorderPlus)
                return;
            }
        }
        else if (pc == PC_INIT_F)
        {
            {
                const RhsSpec sp = specBorderPlus();
                bwdSweepAny<W>(cs, sp);
            }
            phInitPrimalFinish<W>(cs, gp, itp);
            phInitDualRhs<W>(cs, gp, itp);
            {
                const RhsSpec sp = specSingle();
                fwdSweepAny<W>(cs, sp);
                bwdSweepAny<W>(cs, sp);
            }
            phInitDualFinish<W>(cs, gp, itp);
            pc = PC_NORMS;
        }
        else if (pc == PC_NORMS)
        {
            phDataNorms<W>(cs, gp, itp, 0);
            status = -1;
            iter = 0;
            use_backup = false;
            it.bk_valid = 0;
            it.bad = 0;
            inacc_ok = false;
            bk_prev = false;
            pres_prev = 0.;
            pc = PC_HEAD;
        }
        else if (pc == PC_HEAD)
        {
#if IPM_FUSE_UPDATE
            if (iter == 0)
                phResiduals<W, false>(cs, gp, itp);
            else
                phResiduals<W, true>(cs, gp, itp); // applies the step of the previous iteration on its way in (ipm_solve.h)
#else
            phResiduals<W, false>(cs, gp, itp);
#endif
            {
                const double pres = it.pres, dres = it.dres, gap = it.gap;
                const double apc = fabs(it.pcost) > 1e-300 ? fabs(it.pcost) : 1e-300;
                const double relgap = gap / apc;
                const bool nonfinite = !(pres == pres) || !(dres == dres) || !(gap == gap) || fabs(pres) > 1e300 || fabs(dres) > 1e300 || fabs(gap) > 1e300 ||
                                   fabs(it.pcost) > IPM_BLOWN || ipmGapBroken(gap); // a BLOWN-UP iterate is a broken one: see IPM_BLOWN, IPM_NEG_GAP
                if (nonfinite || (bk_prev && iter > 0 && (pres > 500. * pres_prev || gap < 0.)))
                {
                    status = bk_prev ? 0 : -2;
                    use_backup = bk_prev;
                    pc = PC_END;
                    continue;
                }
                pres_prev = pres;
                bk_prev = it.bk_valid != 0;
                if (pres < opt.feastol && dres < opt.feastol && (gap < opt.abstol || relgap < opt.reltol))
                {
                    status = 0;
                    pc = PC_END;
                    continue;
                }
                inacc_ok = pres < 1e-4 && dres < 1e-4 && (gap < 5e-5 || relgap < 5e-5);
                if (iter >= opt.maxit)
                {
                    status = inacc_ok ? 0 : -1;
                    pc = PC_END;
                    continue;
                }
            }
            phScalings<W>(cs, gp, itp);
            if (it.bad)
            {
                status = inacc_ok ? 0 : -2;
                pc = PC_END;
                continue;
            }
            phRhs<W, 0>(cs, gp, itp);
            yield(PC_MAIN_F, (c.ip[IP_SCVX] != 0.) ? 1 : 2);
            return;
        }
        else if (pc == PC_MAIN_F)
        {
            {
                RhsSpec sp;
                sp.n = spec_n;
                bwdSweepAny<W>(cs, sp);
            }
            phDirStage<W, 0>(cs, gp, itp);
            phDirSeg<W, 0>(cs, gp, itp);
            if (!it.bad)
            {
                phRhs<W, 1>(cs, gp, itp);
                const RhsSpec sp = specSingle();
                fwdSweepAny<W>(cs, sp);
                bwdSweepAny<W>(cs, sp);
                phDirStage<W, 1>(cs, gp, itp);
                phDirSeg<W, 1>(cs, gp, itp);
            }
            if (it.bad)
            {
                status = inacc_ok ? 0 : -2;
                pc = PC_END;
                continue;
            }
#if !IPM_FUSE_UPDATE
            phUpdate<W>(cs, gp, itp);
#endif
            iter++; // (IPM_FUSE_UPDATE: the step is applied by the next residual pass, phResiduals<W, true>)
            pc = PC_HEAD;
        }
        else // PC_END: the attempt has ended
        {
            iter_total += iter;
            if (status == 0 || !warm)
                break;
            warm = 0; // a warm start that broke down is repeated from ECOS's cold initialisation
            pc = PC_START;
        }
    }
    iter = iter_total;
    // =============== outputs: readSolution + SC bookkeeping (ipm_kernel's tail, statement for statement) ===============
    KERNEL_TAIL_ARGS(t, a, kernelArgsLate());
    const bool vst = k < K;
    const SV st = makeSV(c.st, L::STREC, unsigned(vst ? k : 0), c.pitch);
    const int fW = use_backup ? int(L::F_WBK) : int(L::F_W);
    double sum_delta = 0.;
    if (vst)
        sum_delta = st[fW + 16];
    sum_delta = wave_sum(sum_delta);
    const double n1 = use_backup ? it.bk_n1 : g.n1, sig = use_backup ? it.bk_sig : g.sig, dsg = use_backup ? it.bk_dsg : g.dsg;
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
            double *Xo = t.X + (size_t(inst) * K + k) * NX, *Uo = t.U + (size_t(inst) * K + k) * NU;
#pragma unroll
            for (int i = 0; i < NX; i++)
                Xo[i] = L::XINV.v[i] >= 0 ? double(st[fW + (L::XINV.v[i] >= 0 ? L::XINV.v[i] : 0)]) : 0.;
#pragma unroll
            for (int i = 0; i < NU; i++)
                Uo[i] = L::UINV.v[i] >= 0 ? double(st[fW + (L::UINV.v[i] >= 0 ? L::UINV.v[i] : 0)]) : 0.;
        }
        if (lane == 0 && c.ip[IP_SCVX] == 0. && c.ip[IP_FIXEDT] == 0.)
            t.sigma[inst] = sig;
    }
    if (lane == 0)
    {
        rl[RL_PC] = double(PC_DONE);
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
                t.status[inst] = status;
                t.active[inst] = 0;
            }
            else
            {
                if (n1 < t.nu_tol)
                    t.wtrx[inst] = it.wtrx * 2.;
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

// ---- kernel F: the factor sweep of every instance that is waiting for one ----
template <class P>
__global__ void __launch_bounds__(WAVE, IPM_WAVES_PER_SIMD) __attribute__((disable_tail_calls)) ipm_factor_kernel(KernelArgs a)
{
    const int inst = blockIdx.x;
    if (inst >= a.B)
        return;
    if (a.active && a.active[inst] == 0)
        return;
    const int lane = threadIdx.x;
    const Ctx c = splitCtx<P>(a, inst, lane);
    const double *rl = c.gsave + GSAVE + RS_LOC;
    const int pc = uniformInt(int(rl[RL_PC]));
    if (pc != PC_INIT_F && pc != PC_MAIN_F)
        return;
    __shared__ TileShared sh;
    __shared__ Ctx cshared;
    const PRIV Ctx *cs = (const PRIV Ctx *)&cshared;
    if (lane == 0)
        cshared = c;
    WAVE_SYNC();
    RhsSpec sp;
    sp.n = uniformInt(int(rl[RL_SPEC]));
    factorSweepAny<P>(cs, sh, sp);
}

} // namespace ipm
} // namespace scpp
