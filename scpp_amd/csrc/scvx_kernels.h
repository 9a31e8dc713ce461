// SCvx mode: batched counterparts of the pieces of the reference's SCvxAlgorithm that sit around the two hot kernels
//   SCvxAlgorithm::solve (cold / warm start)        scpp_core/src/SCvxAlgorithm.cpp:166-227
//   SCvxAlgorithm::getNonlinearCost                 scpp_core/src/SCvxAlgorithm.cpp:262-278  (K-1 nonlinear propagations)
//   accept / reject / radius update of iterate()    scpp_core/src/SCvxAlgorithm.cpp:95-152
// The sub-problem itself (SCvxProblem.cpp:6-71) is solved by ipm_kernel in its SCvx mode (ipm_kernel.h: IP_SCVX).
//
// Streaming engine (continuous batching): the context's B instance SLOTS pull problem instances from a device-side queue.
// A slot whose SCvx loop has terminated hands its result row to the output buffer and is refilled with the next queued
// instance by scvx_stream_refill_kernel at the top of the next round, so every round runs with (nearly) all slots busy
// although the instances need very different numbers of sub-problem solves (mean 26, up to > 60 for the shipped scenario).
#pragma once
#include "discretize_kernel.h"
#include "sc_kernels.h"

namespace scpp
{

constexpr int SCVX_SOLVE_CAP = 64; // sub-problem solves per configured iteration before an instance is retired (see scvxDecide)

struct SCvxBuffers
{
    double *Xold, *Uold;       // candidate backup (td = old_td on rejection); filled by ipm_kernel before it overwrites X / U
    double *tr;                // [B] trust-region radius
    double *last_cost;         // [B] last_nonlinear_cost
    double *cost;              // [B] J of the current candidate
    double *info;              // [B][4]: last rho, actual change, predicted change, accepted code
    int *has_last, *needs_disc, *solves;
    // SCvxAlgorithm::getAllSolutions (SCvxAlgorithm.cpp:192,201,245-260: all_td): opt-in record of the trajectory before the first iteration and
    // after every iteration (= after every ACCEPTED candidate), nondimensional as the algorithm holds it; null = off (the default: the headline's
    // bytes do not move).  Batch entry point only -- a streaming job re-uses a slot for many instances.
    double *iter_ring; // [B][iter_cap][K * (nx + nu) + SCVX_ITER_SCALARS]: X [K][nx], U [K][nu], then {radius, sub-problem solves so far, J, decision code}
    int *iter_count;   // [B] trajectories recorded
    int iter_cap;      // max_iterations + 1
};
constexpr int SCVX_ITER_SCALARS = 4;
__host__ __device__ inline size_t scvxIterRecordDoubles(int K, int nx, int nu) { return size_t(K) * size_t(nx + nu) + SCVX_ITER_SCALARS; }
// the whole wavefront appends the instance's current trajectory to its record (nothing happens when the record is off or full)
__device__ inline void scvxRecordIterate(const SCBuffers &b, const SCvxBuffers &v, long i, int nx, int nu, int lane, int lanes)
{
    if (!v.iter_ring)
        return;
    const int n = v.iter_count[i];
    if (n >= v.iter_cap)
        return;
    const int K = b.K;
    double *dst = v.iter_ring + (size_t(i) * v.iter_cap + n) * scvxIterRecordDoubles(K, nx, nu);
    for (int e = lane; e < K * nx; e += lanes)
        dst[e] = b.X[i * K * nx + e];
    for (int e = lane; e < K * nu; e += lanes)
        dst[K * nx + e] = b.U[i * K * nu + e];
    if (lane == 0)
    {
        double *sc = dst + K * (nx + nu);
        sc[0] = v.tr[i];             // radius after this iteration's update
        sc[1] = double(v.solves[i]); // accepted + rejected candidates so far
        sc[2] = v.last_cost[i];      // nonlinear cost J of the trajectory
        sc[3] = v.info[i * 4 + 3];   // 2 first pass, 1 accepted, 3 converged (0 in the initial record)
    }
    WAVE_SYNC(); // every lane has read the count
    if (lane == 0)
        v.iter_count[i] = n + 1;
}

// per-instance SCvx start-up AFTER scSetupOne (which nondimensionalises, builds the initial or warm trajectory
// and thrust_const): fixed final time, SCvx flags of the sub-problem, radius.
__device__ inline void scvxSetupOne(const SCBuffers &b, const SCvxBuffers &v, const scpp_scvx_opts &so, double final_time, int warm,
                                    long i)
{
    using namespace ipm;
    double *ip = b.ip + i * IP_N;
    ip[IP_SCVX] = 1.;
    ip[IP_WT] = 1.;   // dummy decoupled sigma block (S = 0)
    ip[IP_WTRT] = 1.;
    ip[IP_WTRX] = 0.;
    ip[IP_WVC] = so.weight_virtual_control;
    b.wtrx[i] = 0.;
    if (!warm)
    {
        b.sigma[i] = final_time;
        v.tr[i] = so.trust_region; // loadParameters() on cold start (SCvxAlgorithm.cpp:179)
        v.has_last[i] = 0;
        v.last_cost[i] = 0.;
    }
    ip[IP_TR] = v.tr[i];
    v.needs_disc[i] = 1;
    v.solves[i] = 0;
    v.cost[i] = 0.;
    for (int j = 0; j < 4; j++)
        v.info[i * 4 + j] = 0.;
    b.sc_iters[i] = 1; // iteration++ at the top of the first iterate()
}
__global__ void scvx_setup_kernel(SCBuffers b, SCvxBuffers v, scpp_scvx_opts so, double final_time, int warm)
{
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= b.B)
        return;
    scvxSetupOne(b, v, so, final_time, warm, i);
}
// all_td.push_back(td) before the first iteration (SCvxAlgorithm.cpp:192): one wavefront per instance, after the model's set-up kernel
__global__ void scvx_record_initial_kernel(SCBuffers b, SCvxBuffers v, int nx, int nu)
{
    const long i = blockIdx.x;
    if (i >= b.B || !v.iter_ring)
        return;
    if (threadIdx.x == 0)
        v.iter_count[i] = 0;
    WAVE_SYNC();
    scvxRecordIterate(b, v, i, nx, nu, threadIdx.x, WAVE);
}

// SCvxAlgorithm.cpp:95-152 for one active instance, after the sub-problem solve and the cost evaluation (one lane).
// Returns bit 0: the candidate was rejected (td = old_td: the caller restores X / U from the backup); bit 1: an iteration has ended with td = the
// candidate (first pass, accepted or converged: what SCvxAlgorithm::solve pushes to all_td, SCvxAlgorithm.cpp:201).
__device__ inline int scvxDecide(const SCBuffers &b, const SCvxBuffers &v, const scpp_scvx_opts &so, long i, double nonlinear_cost)
{
    using namespace ipm;
    v.solves[i] += 1;
    if (b.status[i] != 0)
    {
        b.active[i] = 0; // solver failure: the reference terminates (SCvxAlgorithm.cpp:87-91)
        v.needs_disc[i] = 0;
        return 0;
    }
    const double linear_cost = b.norm1_nu[i];
    bool done_iteration = false, converged = false;
    int restore = 0;
    double code = 1.;
    if (!v.has_last[i])
    {
        v.has_last[i] = 1;
        v.last_cost[i] = nonlinear_cost;
        done_iteration = true;
        code = 2.;
    }
    else
    {
        const double actual_change = v.last_cost[i] - nonlinear_cost;
        const double predicted_change = v.last_cost[i] - linear_cost;
        v.last_cost[i] = nonlinear_cost; // (overwritten even when the candidate is rejected, :118)
        v.info[i * 4 + 1] = actual_change;
        v.info[i * 4 + 2] = predicted_change;
        if (fabs(predicted_change) < so.change_threshold)
        {
            converged = true;
            done_iteration = true;
            code = 3.;
        }
        else
        {
            const double rho = actual_change / predicted_change;
            v.info[i * 4 + 0] = rho;
            if (rho < so.rho_0)
            {
                v.tr[i] /= so.alpha;
                restore = 1; // td = old_td ; re-solve without re-discretising
                v.needs_disc[i] = 0;
                code = 0.;
                // The reference's `while (true)` (SCvxAlgorithm.cpp:75-153) has no exit but acceptance.  With the shipped Rocket2D
                // SCvx.info (SI units, radius 5) the radius collapses and some start states are rejected over and over (the
                // candidate creeps, dJ stays < 0): a batched engine cannot spin with them, so an instance is retired with
                // status SCPP_STATUS_REJECTION_CAP once it has used SCVX_SOLVE_CAP x max_iterations sub-problem solves (build-defined).
                if (v.solves[i] >= SCVX_SOLVE_CAP * so.max_iterations)
                {
                    b.status[i] = SCPP_STATUS_REJECTION_CAP;
                    b.active[i] = 0;
                }
            }
            else
            {
                if (rho < so.rho_1)
                    v.tr[i] /= so.alpha;
                else if (rho >= so.rho_2)
                    v.tr[i] *= so.beta;
                done_iteration = true;
            }
        }
    }
    v.info[i * 4 + 3] = code;
    b.ip[i * IP_N + IP_TR] = v.tr[i];
    if (done_iteration)
    {
        if (converged)
        {
            b.converged[i] = 1;
            b.active[i] = 0;
            v.needs_disc[i] = 0;
        }
        else if (b.sc_iters[i] >= so.max_iterations)
        {
            b.active[i] = 0;
            v.needs_disc[i] = 0;
        }
        else
        {
            b.sc_iters[i] += 1;
            v.needs_disc[i] = 1;
        }
    }
    return restore | (done_iteration ? 2 : 0);
}

// getNonlinearCost + the accept / reject / radius logic: one wavefront per instance.  Lane k propagates segment k with the
// nonlinear dynamics (RKF78 x 20 like scpp::simulate) and contributes ||x_prop - x_{k+1}||_1; fixed summation order
// (wave_sum) keeps the accept/reject decisions reproducible.  Lane 0 then takes the decision of iterate(); a rejected
// candidate is rolled back by the whole wavefront (td = old_td).
#ifndef COST_WAVES_PER_SIMD
#define COST_WAVES_PER_SIMD 1
#endif
template <class Model>
__device__ __forceinline__ void scvxCostUpdate(const SCBuffers &b, const SCvxBuffers &v, const scpp_scvx_opts &so, const long i)
{
    using namespace ipm;
    constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP;
    if (i >= b.B || b.active[i] == 0)
        return;
    const int K = b.K, k = threadIdx.x;
    const int foh = so.interpolate_input;
    double acc = 0.;
    if (k < K - 1 && b.status[i] == 0)
    {
        double p[NP], u0[NU], u1[NU], y[NX], kk[RK_S][NX];
        const double *ip = b.ip + i * IP_N;
        for (int j = 0; j < NP; j++)
            p[j] = ip[IP_PAR + j];
        const double *X = b.X + (i * K + k) * NX, *U = b.U + (i * K + k) * NU;
        for (int j = 0; j < NU; j++)
        {
            u0[j] = U[j];
            u1[j] = foh ? U[NU + j] : U[j];
        }
        for (int j = 0; j < NX; j++)
            y[j] = X[j];
        const double dt = b.sigma[i] / double(K - 1);
        const double h = dt / 20.;
        for (int step = 0; step < 20; step++)
        {
            const double t0 = double(step) * h;
#pragma unroll
            for (int s = 0; s < RK_S; s++)
            {
                double ys[NX], u[NU];
                const double ts = t0 + RK_C[s] * h;
                for (int j = 0; j < NX; j++)
                {
                    double a = 0.;
#pragma unroll
                    for (int q = 0; q < s; q++)
                        if (RK_A[s][q] != 0.)
                            a += RK_A[s][q] * kk[q][j];
                    ys[j] = y[j] + h * a;
                }
                for (int j = 0; j < NU; j++)
                    u[j] = u0[j] + ts / dt * (u1[j] - u0[j]);
                Model::template systemFlowMap<double>(ys, u, p, kk[s]);
            }
            for (int j = 0; j < NX; j++)
            {
                double a = 0.;
#pragma unroll
                for (int s = 0; s < RK_S; s++)
                    if (RK_B[s] != 0.)
                        a += RK_B[s] * kk[s][j];
                y[j] += h * a;
            }
        }
        for (int j = 0; j < NX; j++)
            acc += fabs(y[j] - X[NX + j]);
    }
    acc = wave_sum(acc);
    int flags = 0;
    if (k == 0)
    {
        v.cost[i] = acc;
        flags = scvxDecide(b, v, so, i, acc);
    }
    flags = __shfl(flags, 0);
    if (flags & 1)
    {
        for (int e = k; e < K * NX; e += WAVE)
            b.X[i * K * NX + e] = v.Xold[i * K * NX + e];
        for (int e = k; e < K * NU; e += WAVE)
            b.U[i * K * NU + e] = v.Uold[i * K * NU + e];
    }
    if (flags & 2)
        scvxRecordIterate(b, v, i, NX, NU, k, WAVE);
}
template <class Model>
__global__ void __launch_bounds__(WAVE, COST_WAVES_PER_SIMD) scvx_cost_update_kernel(SCBuffers b, SCvxBuffers v, scpp_scvx_opts so)
{
    scvxCostUpdate<Model>(b, v, so, blockIdx.x);
}

// The same step for a kernel with a 256-register budget (the persistent SCvx kernel, scvx_persistent.h).  scvxCostUpdate gives a lane one
// whole segment: 13 stage slopes of NX doubles = 364 VGPRs for RocketQuat -- fine at one wavefront per SIMD, but inside the persistent kernel
// they went to scratch memory and one candidate cost 2.0 M cycles against 0.25 M stand-alone (measured, round 5).  Here TWO lanes share a
// segment: lane (pair p, half h) keeps half h of the state and of every stage slope (91 doubles), both evaluate the whole flow map on the stage
// value they assemble with one exchange between neighbours per stage, and the K - 1 segments go through in two passes of up to 32 pairs.
// BITWISE the result of scvxCostUpdate: every component goes through the same operations in the same order (the stage combination and the
// step update are per component; the flow map sees the same stage value), the segment's defect sum continues from half 0 to half 1 in component
// order, and the per-segment sums are reduced in the lane = segment arrangement the one-lane-per-segment code has (through `seg_sum`, LDS).
template <class Model>
__device__ __forceinline__ void scvxCostUpdateSplit(const SCBuffers &b, const SCvxBuffers &v, const scpp_scvx_opts &so, const long i, double *seg_sum /* [64] LDS */)
{
    using namespace ipm;
    // half 0 keeps components [0, NH), half 1 components [NH, NX): NH = NX / 2 for an even number of states; for an odd one (round 6: Lander3dof
    // has seven) half 1 holds one component less and its last slot is padding that is carried as 0 and never stored or summed
    constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP, NH = (NX + 1) / 2, NH1 = NX - NH;
    if (i >= b.B || b.active[i] == 0)
        return;
    const int K = b.K, lane = threadIdx.x;
    const int foh = so.interpolate_input;
    const int half = lane & 1, pair = lane >> 1;
    const bool solved = b.status[i] == 0;
    if (lane < WAVE)
        seg_sum[lane] = 0.;
    WAVE_SYNC();
    for (int pass = 0; pass * 32 < K - 1; pass++)
    {
        const int k = pass * 32 + pair; // this pair's segment
        const bool on = k < K - 1 && solved;
        const int kc = on ? k : 0;
        double p[NP], u0[NU], u1[NU], y[NH], kk[RK_S][NH];
        const double *ip = b.ip + i * IP_N;
#pragma unroll
        for (int j = 0; j < NP; j++)
            p[j] = ip[IP_PAR + j];
        const double *X = b.X + (i * K + kc) * NX, *U = b.U + (i * K + kc) * NU;
#pragma unroll
        for (int j = 0; j < NU; j++)
        {
            u0[j] = U[j];
            u1[j] = foh ? U[NU + j] : U[j];
        }
#pragma unroll
        for (int j = 0; j < NH; j++)
            y[j] = (j < NH1 || !half) ? X[half * NH + j] : 0.;
        const double dt = b.sigma[i] / double(K - 1);
        const double h = dt / 20.;
        for (int step = 0; step < 20; step++)
        {
            const double t0 = double(step) * h;
#pragma unroll
            for (int s = 0; s < RK_S; s++)
            {
                double mine[NH], ys[NX], u[NU], f[NX];
                const double ts = t0 + RK_C[s] * h;
#pragma unroll
                for (int j = 0; j < NH; j++)
                {
                    double a = 0.;
#pragma unroll
                    for (int q = 0; q < s; q++)
                        if (RK_A[s][q] != 0.)
                            a += RK_A[s][q] * kk[q][j];
                    mine[j] = y[j] + h * a;
                }
                // the other half of the stage value from the neighbour lane
#pragma unroll
                for (int j = 0; j < NH; j++)
                {
                    const double other = __shfl_xor(mine[j], 1);
                    ys[j] = half ? other : mine[j];
                    if (j < NH1)
                        ys[NH + j] = half ? mine[j] : other;
                }
#pragma unroll
                for (int j = 0; j < NU; j++)
                    u[j] = u0[j] + ts / dt * (u1[j] - u0[j]);
                Model::template systemFlowMap<double>(ys, u, p, f);
#pragma unroll
                for (int j = 0; j < NH; j++)
                    kk[s][j] = half ? (j < NH1 ? f[NH + j < NX ? NH + j : 0] : 0.) : f[j];
            }
#pragma unroll
            for (int j = 0; j < NH; j++)
            {
                double a = 0.;
#pragma unroll
                for (int s = 0; s < RK_S; s++)
                    if (RK_B[s] != 0.)
                        a += RK_B[s] * kk[s][j];
                y[j] += h * a;
            }
        }
        // defect of the segment: components 0 .. NX-1 in order -- half 0 sums its components, half 1 continues from that partial sum
        double part = 0.;
        if (!half)
        {
#pragma unroll
            for (int j = 0; j < NH; j++)
                part += fabs(y[j] - X[NX + j]);
        }
        const double from0 = __shfl_xor(part, 1);
        if (half)
        {
            part = from0;
#pragma unroll
            for (int j = 0; j < NH1; j++)
                part += fabs(y[j] - X[NX + NH + j]);
            if (on)
                seg_sum[k] = part;
        }
    }
    WAVE_SYNC();
    double acc = (lane < K - 1) ? seg_sum[lane] : 0.; // lane = segment, as in scvxCostUpdate: the same reduction order
    acc = wave_sum(acc);
    WAVE_SYNC();
    int flags = 0;
    if (lane == 0)
    {
        v.cost[i] = acc;
        flags = scvxDecide(b, v, so, i, acc);
    }
    flags = __shfl(flags, 0);
    if (flags & 1)
    {
        for (int e = lane; e < K * NX; e += WAVE)
            b.X[i * K * NX + e] = v.Xold[i * K * NX + e];
        for (int e = lane; e < K * NU; e += WAVE)
            b.U[i * K * NU + e] = v.Uold[i * K * NU + e];
    }
    if (flags & 2)
        scvxRecordIterate(b, v, i, NX, NU, lane, WAVE);
}

// ---------------------------------------------------------------- streaming engine
// result row of one instance (doubles): X [K][nx], U [K][nu] (dimensional), then the scalars below
enum StreamRow
{
    SR_SIGMA = 0,
    SR_NU,     // final ||nu||_1 (linear cost of the last sub-problem)
    SR_COST,   // last nonlinear cost J
    SR_TR,     // final trust-region radius
    SR_ITERS,  // SCvx iterations
    SR_SOLVES, // sub-problem solves
    SR_CONV,   // 1: |dL| < change_threshold (SCvxAlgorithm.cpp:125)
    SR_STATUS, // 0 ok, < 0 interior-point failure
    SR_IPM,    // interior-point iterations over all solves
    SR_INST,   // instance id (row index): self-check of the hand-over
    SR_NSCALARS
};
__host__ __device__ inline int streamRowDoubles(int K, int nx, int nu) { return K * (nx + nu) + SR_NSCALARS; }

struct StreamQueue
{
    int N;                 // instances in the job
    const double *x_init;  // [N][nx] dimensional initial states
    double *rows;          // [N][streamRowDoubles]
    int *head;             // next instance id to hand out
    int *done;             // instances whose row has been written
    int *nconv;            // converged instances
    int *slot_inst;        // [B] instance id held by the slot, -1: empty
    int *warm;             // [B] interior-point warm-start flag of the slot
};

// What the refill kernel needs to know about a model is part of its plugin (sc_kernels.h): the C-ABI parameter struct, the cold start
// of one instance spread over a wavefront's lanes (setupOne with k0 = lane, kstep = 64) and the factors that redimensionalise a result row.
using RefillRocketQuat = RocketQuatPlugin;
using RefillRocket2d = Rocket2dPlugin;

// One wavefront per slot, at the top of every round: harvest a terminated instance (redimensionalised row -> rows[inst]),
// then pull the next instance id off the queue and run its cold start (the model's set-up + scvxSetupOne).
template <class T>
__device__ __forceinline__ void scvxStreamRefill(const SCBuffers &b, const SCvxBuffers &v, const StreamQueue &q, const typename T::Params &mp,
                                                 const scpp_sc_opts &sc, const scpp_scvx_opts &so, const long slot)
{
    using namespace ipm;
    constexpr int NX = T::NX, NU = T::NU;
    if (slot >= b.B || b.active[slot] != 0)
        return;
    const int K = b.K, lane = threadIdx.x;
    const int inst = q.slot_inst[slot];
    if (inst >= 0)
    {
        const double *ip = b.ip + slot * IP_N;
        const double ms = so.nondimensionalize ? ip[IP_MSCALE] : 1., rs = so.nondimensionalize ? ip[IP_RSCALE] : 1.;
        double *row = q.rows + size_t(inst) * streamRowDoubles(K, NX, NU);
        for (int e = lane; e < K * NX; e += WAVE)
            row[e] = b.X[slot * K * NX + e] * T::rx(e % NX, ms, rs);
        for (int e = lane; e < K * NU; e += WAVE)
            row[K * NX + e] = b.U[slot * K * NU + e] * T::ru(e % NU, ms, rs);
        if (lane == 0)
        {
            double *s = row + K * (NX + NU);
            s[SR_SIGMA] = b.sigma[slot];
            s[SR_NU] = b.norm1_nu[slot];
            s[SR_COST] = v.last_cost[slot];
            s[SR_TR] = v.tr[slot];
            s[SR_ITERS] = double(b.sc_iters[slot]);
            s[SR_SOLVES] = double(v.solves[slot]);
            s[SR_CONV] = double(b.converged[slot]);
            s[SR_STATUS] = double(b.status[slot]);
            s[SR_IPM] = double(b.ipm_iters[slot]);
            s[SR_INST] = double(inst);
            if (b.converged[slot])
                atomicAdd(q.nconv, 1);
        }
    }
    WAVE_SYNC(); // every lane has read the slot's old instance id and state before lane 0 replaces them
    int next = -1;
    if (lane == 0)
    {
        if (*(volatile int *)q.head < q.N)
        {
            next = atomicAdd(q.head, 1);
            if (next >= q.N)
                next = -1;
        }
        q.slot_inst[slot] = next;
    }
    next = __shfl(next, 0);
    if (next >= 0)
    {
        const double *xi = q.x_init + size_t(next) * NX;
        T::setupOne(b, mp, sc, 0, slot, xi, lane, WAVE);
        if (lane == 0)
        {
            scvxSetupOne(b, v, so, mp.final_time, 0, slot);
            q.warm[slot] = 0; // a new instance starts from ECOS's cold initialisation
        }
    }
    if (lane == 0 && inst >= 0)
    {
        __threadfence();
        atomicAdd(q.done, 1); // after the row is complete
    }
}
template <class T>
__global__ void __launch_bounds__(WAVE) scvx_stream_refill_kernel(SCBuffers b, SCvxBuffers v, StreamQueue q, typename T::Params mp,
                                                                   scpp_sc_opts sc, scpp_scvx_opts so)
{
    scvxStreamRefill<T>(b, v, q, mp, sc, so, blockIdx.x);
}

} // namespace scpp
