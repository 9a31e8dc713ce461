// SCvx mode: batched counterparts of the pieces of the reference's SCvxAlgorithm that sit around the two hot kernels
//   SCvxAlgorithm::solve (cold / warm start)        scpp_core/src/SCvxAlgorithm.cpp:166-227
//   SCvxAlgorithm::getNonlinearCost                 scpp_core/src/SCvxAlgorithm.cpp:262-278  (K-1 nonlinear propagations)
//   accept / reject / radius update of iterate()    scpp_core/src/SCvxAlgorithm.cpp:95-152
// The sub-problem itself (SCvxProblem.cpp:6-71) is solved by ipm_kernel in its SCvx mode (ipm_kernel.h: IP_SCVX).
#pragma once
#include "discretize_kernel.h"
#include "sc_kernels.h"

namespace scpp
{

struct SCvxBuffers
{
    double *Xold, *Uold;       // candidate backup (td = old_td on rejection)
    double *tr;                // [B] trust-region radius
    double *last_cost;         // [B] last_nonlinear_cost
    double *cost;              // [B] J of the current candidate
    double *info;              // [B][4]: last rho, actual change, predicted change, accepted code
    int *has_last, *needs_disc, *solves;
};

// per-instance SCvx start-up AFTER sc_setup_kernel (which nondimensionalises, builds the initial or warm trajectory
// and thrust_const): fixed final time, SCvx flags of the sub-problem, radius.
__global__ void scvx_setup_kernel(SCBuffers b, SCvxBuffers v, scpp_scvx_opts so, double final_time, int warm)
{
    using namespace ipm;
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= b.B)
        return;
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

// getNonlinearCost: one wavefront per instance, lane k propagates segment k with the nonlinear dynamics (RKF78 x 20
// like scpp::simulate) and contributes ||x_prop - x_{k+1}||_1; fixed summation order (wave_sum) keeps the
// accept/reject decisions reproducible.
template <class Model>
__global__ void __launch_bounds__(WAVE) scvx_cost_kernel(SCBuffers b, SCvxBuffers v, int foh)
{
    using namespace ipm;
    constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP;
    const long i = blockIdx.x;
    if (i >= b.B || b.active[i] == 0)
        return;
    const int K = b.K, k = threadIdx.x;
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
    if (k == 0)
        v.cost[i] = acc;
}

// SCvxAlgorithm.cpp:95-152 for every active instance, after the sub-problem solve and the cost evaluation
__global__ void scvx_update_kernel(SCBuffers b, SCvxBuffers v, scpp_scvx_opts so)
{
    using namespace ipm;
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= b.B || b.active[i] == 0)
        return;
    v.solves[i] += 1;
    if (b.status[i] != 0)
    {
        b.active[i] = 0; // solver failure: the reference terminates (SCvxAlgorithm.cpp:87-91)
        v.needs_disc[i] = 0;
        return;
    }
    const int K = b.K;
    const double nonlinear_cost = v.cost[i], linear_cost = b.norm1_nu[i];
    bool done_iteration = false, converged = false;
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
                // td = old_td ; re-solve without re-discretising
                for (int e = 0; e < K * 14; e++)
                    b.X[i * K * 14 + e] = v.Xold[i * K * 14 + e];
                for (int e = 0; e < K * 4; e++)
                    b.U[i * K * 4 + e] = v.Uold[i * K * 4 + e];
                v.needs_disc[i] = 0;
                code = 0.;
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
}

} // namespace scpp
