// Element-wise kernels around the two hot kernels: per-instance problem setup, trajectory
// initialisation and (re)dimensionalisation -- batched counterparts of
//   RocketQuat::Parameters::nondimensionalize / redimensionalize   scpp_models/src/rocketQuat.cpp:291-332
//   RocketQuat::getInitializedTrajectory                           scpp_models/src/rocketQuat.cpp:39-68
//   RocketQuat::getNewModelParameters / updateProblemParameters    scpp_models/src/rocketQuat.cpp:156-173
//   RocketQuat::(non|re)dimensionalizeTrajectory                    scpp_models/src/rocketQuat.cpp:175-201
//   SCAlgorithm::solve (cold / warm start bookkeeping)              scpp_core/src/SCAlgorithm.cpp:134-160,182-187
// One thread per instance (K <= 64 nodes each; negligible next to the solver).
#pragma once
#include "ipm_kernel.h"
#include "model_rocketquat.h"
#include "model_lander3dof.h"
#include "../../include/scpp_hip.h"

namespace scpp
{

struct SCBuffers
{
    int B, K;
    const double *x_init_dim; // [B][14] dimensional initial states
    double *X, *U, *sigma;    // trajectory (nondimensional while solving)
    double *ip;               // [B][IP_N]
    double *uhat;             // [B][K][3]
    double *wtrx;
    int *active, *converged, *sc_iters, *ipm_iters, *status;
    double *norm1_nu, *sum_delta, *delta_sigma;
};

// cold (warm=0) or warm (warm=1) start of SCAlgorithm::solve of ONE instance (slot i), written so that it can run on one
// thread (k0 = 0, kstep = 1: sc_setup_kernel) or spread over the lanes of a wavefront (k0 = lane, kstep = 64: the
// streaming engine refills a finished slot with the next queued instance).  xi: the instance's dimensional initial state.
__device__ inline void scSetupOne(const SCBuffers &b, const scpp_rocketquat_params &mp, const scpp_sc_opts &so, int warm,
                                  long i, const double *xi, int k0, int kstep)
{
    using namespace ipm;
    const int K = b.K;
    double *ip = b.ip + i * IP_N;
    double m_scale = 1., r_scale = 1.;
    if (so.nondimensionalize)
    {
        m_scale = xi[0];
        r_scale = sqrt(xi[1] * xi[1] + xi[2] * xi[2] + xi[3] * xi[3]);
    }
    double x0[14], xf[14];
    for (int j = 0; j < 14; j++)
    {
        x0[j] = xi[j];
        xf[j] = mp.x_final[j];
    }
    x0[0] /= m_scale;
    xf[0] /= m_scale;
    for (int j = 1; j < 7; j++)
    {
        x0[j] /= r_scale;
        xf[j] /= r_scale;
    }
    const double T_min = mp.T_min / (m_scale * r_scale), T_max = mp.T_max / (m_scale * r_scale);
    if (k0 == 0)
    {
        for (int j = 0; j < 14; j++)
        {
            ip[IP_XINIT + j] = x0[j];
            ip[IP_XFINAL + j] = xf[j];
        }
        ip[IP_GS] = tan(mp.gamma_gs);
        ip[IP_TILT] = sqrt((1. - cos(mp.theta_max)) / 2.);
        ip[IP_WMAX] = mp.w_B_max;
        ip[IP_TMIN] = T_min;
        ip[IP_TMAX] = T_max;
        ip[IP_GIM] = tan(mp.gimbal_max);
        ip[IP_MDRY] = xf[0];
        // free_final_time false (SCProblem.cpp:33-35,78-100): no sigma / delta_sigma in the reference's problem.  Here sigma stays
        // in the structure as a DECOUPLED dummy block (dS/dsigma = 0 from the fixed-time discretisation, unit weights), exactly
        // as in SCvx mode, and the solver does not write it back: the final time stays at its configured value.
        ip[IP_WT] = so.free_final_time ? so.weight_time : 1.;
        ip[IP_WTRT] = so.free_final_time ? so.weight_trust_region_time : 1.;
        ip[IP_WTRX] = so.weight_trust_region_trajectory;
        ip[IP_WVC] = so.weight_virtual_control;
        ip[IP_FIXEDT] = so.free_final_time ? 0. : 1.;
        ip[IP_PAR + 0] = mp.alpha_m * r_scale;
        for (int j = 0; j < 3; j++)
        {
            ip[IP_PAR + 1 + j] = mp.g_I[j] / r_scale;
            ip[IP_PAR + 4 + j] = mp.J_B[j] / (m_scale * r_scale * r_scale);
            ip[IP_PAR + 7 + j] = mp.r_T_B[j] / r_scale;
        }
        ip[IP_MSCALE] = m_scale;
        ip[IP_RSCALE] = r_scale;
        ip[IP_FINALTIME] = mp.final_time;
        ip[IP_SCVX] = 0.;
        ip[IP_TR] = 0.;
    }

    double *X = b.X + i * K * 14, *U = b.U + i * K * 4;
    for (int k = k0; k < K; k += kstep)
    {
        double *x = X + k * 14, *u = U + k * 4;
        if (!warm)
        {
            // getInitializedTrajectory (k/K interpolation quirk kept)
            const double a1 = double(K - k) / K, a2 = double(k) / K;
            for (int j = 0; j < 7; j++)
                x[j] = a1 * x0[j] + a2 * xf[j];
            const double *q0 = x0 + 7, *q1 = xf + 7;
            const double one = 1. - 2.220446049250313e-16;
            const double d = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
            const double absD = fabs(d);
            double s0, s1;
            if (absD >= one)
            {
                s0 = 1. - a2;
                s1 = a2;
            }
            else
            {
                const double th = acos(absD), st = sin(th);
                s0 = sin((1. - a2) * th) / st;
                s1 = sin(a2 * th) / st;
            }
            if (d < 0.)
                s1 = -s1;
            for (int j = 0; j < 4; j++)
                x[7 + j] = s0 * q0[j] + s1 * q1[j];
            for (int j = 11; j < 14; j++)
                x[j] = a1 * x0[j] + a2 * xf[j];
            u[0] = 0.;
            u[1] = 0.;
            u[2] = (T_max - T_min) / 2.;
            u[3] = 0.;
        }
        else
        {
            // warm start: stored trajectory is dimensional -> nondimensionalizeTrajectory
            // (weight_trust_region_trajectory keeps its doubled value: loadParameters() is skipped)
            x[0] /= m_scale;
            for (int j = 1; j < 7; j++)
                x[j] /= r_scale;
            for (int j = 0; j < 3; j++)
                u[j] /= m_scale * r_scale;
            u[3] /= m_scale * r_scale * r_scale;
        }
        if (!so.interpolate_input && k == K - 1) // zero-order hold: K-1 inputs (trajectoryData.hpp:27-32); the slot of node K-1 is unused
            u[0] = u[1] = u[2] = u[3] = 0.;
        // updateProblemParameters: thrust_const from the trajectory bound at solve() start
        double *uh = b.uhat + (i * K + k) * 3;
        if (mp.exact_minimum_thrust)
        {
            const double z = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
            const double s = z > 0. ? 1. / sqrt(z) : 1.;
            for (int j = 0; j < 3; j++)
                uh[j] = u[j] * s;
        }
        else
        {
            uh[0] = 0.;
            uh[1] = 0.;
            uh[2] = 1.;
        }
    }
    if (k0 == 0)
    {
        if (!warm)
        {
            b.sigma[i] = mp.final_time;
            b.wtrx[i] = so.weight_trust_region_trajectory; // loadParameters() on cold start
        }
        b.active[i] = 1;
        b.converged[i] = 0;
        b.sc_iters[i] = 0;
        b.ipm_iters[i] = 0;
        b.status[i] = 0;
        b.norm1_nu[i] = 0.;
        b.sum_delta[i] = 0.;
        b.delta_sigma[i] = 0.;
    }
}

// ---- Rocket2d: Parameters::nondimensionalize (rocket2d.cpp:200-216), getInitializedTrajectory (:120-135),
//      getNewModelParameters (:143-148), (non|re)dimensionalizeTrajectory (:96-118).  One thread per instance. ----
__device__ inline void scSetupOneR2d(const SCBuffers &b, const scpp_rocket2d_params &mp, const scpp_sc_opts &so, int warm, long i,
                                     const double *xi)
{
    using namespace ipm;
    const int K = b.K;
    double *ip = b.ip + i * IP_N;
    double m_scale = 1., r_scale = 1.;
    if (so.nondimensionalize)
    {
        r_scale = sqrt(xi[0] * xi[0] + xi[1] * xi[1]);
        m_scale = mp.m;
    }
    double x0[6], xf[6];
    for (int j = 0; j < 6; j++)
    {
        x0[j] = xi[j];
        xf[j] = mp.x_final[j];
    }
    for (int j = 0; j < 4; j++)
    {
        x0[j] /= r_scale;
        xf[j] /= r_scale;
    }
    for (int j = 0; j < IP_N; j++)
        ip[j] = 0.;
    for (int j = 0; j < 6; j++)
    {
        ip[IP_XINIT + j] = x0[j];
        ip[IP_XFINAL + j] = xf[j];
    }
    const double T_min = mp.T_min / (m_scale * r_scale), T_max = mp.T_max / (m_scale * r_scale);
    ip[IP_GS] = tan(mp.gamma_gs);
    ip[IP_TILT] = mp.theta_max;
    ip[IP_WMAX] = mp.w_B_max;
    ip[IP_TMIN] = T_min;
    ip[IP_TMAX] = T_max;
    ip[IP_GIM] = mp.gimbal_max;
    ip[IP_WT] = so.free_final_time ? so.weight_time : 1.; // fixed final time: decoupled dummy sigma block (see scSetupOne)
    ip[IP_WTRT] = so.free_final_time ? so.weight_trust_region_time : 1.;
    ip[IP_WTRX] = so.weight_trust_region_trajectory;
    ip[IP_WVC] = so.weight_virtual_control;
    ip[IP_FIXEDT] = so.free_final_time ? 0. : 1.;
    ip[IP_PAR + 0] = mp.m / m_scale;
    ip[IP_PAR + 1] = mp.J_B / (m_scale * r_scale * r_scale);
    ip[IP_PAR + 2] = mp.g_I[0] / r_scale;
    ip[IP_PAR + 3] = mp.g_I[1] / r_scale;
    ip[IP_PAR + 4] = mp.r_T_B[0] / r_scale;
    ip[IP_PAR + 5] = mp.r_T_B[1] / r_scale;
    ip[IP_MSCALE] = m_scale;
    ip[IP_RSCALE] = r_scale;
    ip[IP_FINALTIME] = mp.final_time;
    double *X = b.X + i * K * 6, *U = b.U + i * K * 2;
    for (int k = 0; k < K; k++)
    {
        double *x = X + k * 6, *u = U + k * 2;
        if (!warm)
        {
            const double a1 = double(K - k) / K, a2 = double(k) / K;
            for (int j = 0; j < 6; j++)
                x[j] = a1 * x0[j] + a2 * xf[j];
            u[0] = 0.;
            u[1] = (T_max + T_min) / 2.;
        }
        else
        {
            for (int j = 0; j < 4; j++)
                x[j] /= r_scale;
            u[1] /= m_scale * r_scale;
        }
        if (!so.interpolate_input && k == K - 1) // zero-order hold: the input slot of node K-1 is unused
            u[0] = u[1] = 0.;
        double *uh = b.uhat + (i * K + k) * 3; // no linearised-thrust row in this model's table
        uh[0] = 0.;
        uh[1] = 0.;
        uh[2] = 1.;
    }
    if (!warm)
    {
        b.sigma[i] = mp.final_time;
        b.wtrx[i] = so.weight_trust_region_trajectory;
    }
    b.active[i] = 1;
    b.converged[i] = 0;
    b.sc_iters[i] = 0;
    b.ipm_iters[i] = 0;
    b.status[i] = 0;
    b.norm1_nu[i] = 0.;
    b.sum_delta[i] = 0.;
    b.delta_sigma[i] = 0.;
}
// factor that redimensionalises entry j of a state (input) vector
__host__ __device__ inline double redimX(int j, double m_scale, double r_scale) { return j == 0 ? m_scale : (j < 7 ? r_scale : 1.); }
__host__ __device__ inline double redimU(int j, double m_scale, double r_scale)
{
    return j < 3 ? m_scale * r_scale : m_scale * r_scale * r_scale;
}

// ------------------------------------------------------------------------------------------------------------
// MODEL PLUGINS (round 6): everything the engine knows about a model, in ONE struct -- the counterpart of a class derived from
// SystemModel (scpp_core/include/systemModel.hpp:13-159).  A plugin names
//   Model         the flow map (model_*.h: systemFlowMap<T>; systemModel.hpp:69-73)
//   Table         the application constraints (constraint_table.h; addApplicationConstraints, systemModel.hpp:76-82)
//   Params        the model's C-ABI parameter struct (include/scpp_hip.h; the reference's Model::Parameters)
//   setupOne      nondimensionalize + getInitializedTrajectory + getNewModelParameters + the dynpar block of one instance, written so that
//                 it runs on one thread (k0 = 0, kstep = 1) or spread over a wavefront's lanes (k0 = lane, kstep = 64)
//   rx / ru       redimensionalizeTrajectory as one factor per state / input entry
//   supported     what of the parameter struct this engine refuses (SCPP_E_UNSUPPORTED)
// and the list at the end of this file (`Plugins`) is the ONLY place a model is registered: csrc/scpp_hip.cpp dispatches every launch
// through withPlugin(model_id, ...), the set-up / redimensionalisation / refill / persistent kernels are templates over the plugin.
// A new model = its flow map + its table + one plugin struct + one entry in the list (+ its parameter struct and entry points in the C ABI).
// ------------------------------------------------------------------------------------------------------------
// the solver table of the persistent kernel: its own type, so that its phase functions are instantiated for this kernel's register budget only
struct PersistRocketQuat : ipm::RocketQuatSC
{
};
struct PersistRocket2d : ipm::Rocket2dSC
{
};

struct RocketQuatPlugin
{
    static constexpr int ID = SCPP_MODEL_ROCKETQUAT;
    using Model = RocketQuatModel;
    using Table = ipm::RocketQuatSC;
    using PersistTable = PersistRocketQuat;
    using Params = scpp_rocketquat_params;
    static constexpr int NX = Model::NX, NU = Model::NU;
    // kernels only this model is instantiated for: the split schedule of the interior-point solve (ipm_split.h) and the persistent SC kernel
    static constexpr bool SPLIT_SCHEDULE = true, SC_PERSISTENT = true, SCVX_PERSISTENT = true;
    // enable_roll_control = true (rocketQuat.cpp:135-138): 18 free variables per node do not fit the 16-wide tile
    static bool supported(const Params &mp) { return !mp.enable_roll_control; }
    static __device__ void setupOne(const SCBuffers &b, const Params &mp, const scpp_sc_opts &so, int warm, long i, const double *xi, int k0, int kstep)
    {
        scSetupOne(b, mp, so, warm, i, xi, k0, kstep);
    }
    static __host__ __device__ double rx(int j, double ms, double rs) { return redimX(j, ms, rs); } // rocketQuat.cpp:188-201
    static __host__ __device__ double ru(int j, double ms, double rs) { return redimU(j, ms, rs); }
};

struct Rocket2dPlugin
{
    static constexpr int ID = SCPP_MODEL_ROCKET2D;
    using Model = Rocket2dModel;
    using Table = ipm::Rocket2dSC;
    using PersistTable = PersistRocket2d;
    using Params = scpp_rocket2d_params;
    static constexpr int NX = Model::NX, NU = Model::NU;
    static constexpr bool SPLIT_SCHEDULE = false, SC_PERSISTENT = false, SCVX_PERSISTENT = true;
    static bool supported(const Params &) { return true; }
    static __device__ void setupOne(const SCBuffers &b, const Params &mp, const scpp_sc_opts &so, int warm, long i, const double *xi, int k0, int)
    {
        if (k0 == 0) // (one thread per instance: K <= 64 nodes of 8 numbers)
            scSetupOneR2d(b, mp, so, warm, i, xi);
    }
    static __host__ __device__ double rx(int j, double, double rs) { return j < 4 ? rs : 1.; } // rocket2d.cpp:108-118
    static __host__ __device__ double ru(int j, double ms, double rs) { return j == 1 ? ms * rs : 1.; }
};

// ---- Lander3dof (csrc/model_lander3dof.h; not a model of the reference): the whole of what a third model adds to this file ----
// nondimensionalize (m_scale = m_init, r_scale = |r_init|, as RocketQuat), getInitializedTrajectory (the models' k / K interpolation, hover-ish
// thrust), getNewModelParameters (flow-map parameters, thrust_const of the linearised minimum thrust) and the dynpar block of one instance.
__device__ inline void scSetupOneLander(const SCBuffers &b, const scpp_lander3dof_params &mp, const scpp_sc_opts &so, int warm, long i, const double *xi,
                                        int k0, int kstep)
{
    using namespace ipm;
    const int K = b.K;
    double *ip = b.ip + i * IP_N;
    double m_scale = 1., r_scale = 1.;
    if (so.nondimensionalize)
    {
        m_scale = xi[0];
        r_scale = sqrt(xi[1] * xi[1] + xi[2] * xi[2] + xi[3] * xi[3]);
    }
    double x0[7], xf[7];
    for (int j = 0; j < 7; j++)
    {
        x0[j] = xi[j] / (j == 0 ? m_scale : r_scale);
        xf[j] = mp.x_final[j] / (j == 0 ? m_scale : r_scale);
    }
    const double T_min = mp.T_min / (m_scale * r_scale), T_max = mp.T_max / (m_scale * r_scale);
    if (k0 == 0)
    {
        for (int j = 0; j < IP_N; j++)
            ip[j] = 0.;
        for (int j = 0; j < 7; j++)
        {
            ip[IP_XINIT + j] = x0[j];
            ip[IP_XFINAL + j] = xf[j];
        }
        ip[IP_GS] = tan(mp.gamma_gs);
        ip[IP_TMIN] = T_min;
        ip[IP_TMAX] = T_max;
        ip[IP_GIM] = tan(mp.pointing_max);
        ip[IP_MDRY] = xf[0];
        ip[IP_WT] = so.free_final_time ? so.weight_time : 1.; // fixed final time: decoupled dummy sigma block (see scSetupOne)
        ip[IP_WTRT] = so.free_final_time ? so.weight_trust_region_time : 1.;
        ip[IP_WTRX] = so.weight_trust_region_trajectory;
        ip[IP_WVC] = so.weight_virtual_control;
        ip[IP_FIXEDT] = so.free_final_time ? 0. : 1.;
        ip[IP_PAR + 0] = mp.alpha_m * r_scale;
        for (int j = 0; j < 3; j++)
            ip[IP_PAR + 1 + j] = mp.g_I[j] / r_scale;
        ip[IP_MSCALE] = m_scale;
        ip[IP_RSCALE] = r_scale;
        ip[IP_FINALTIME] = mp.final_time;
    }
    for (int k = k0; k < K; k += kstep)
    {
        double *x = b.X + (i * K + k) * 7, *u = b.U + (i * K + k) * 3;
        if (!warm)
        {
            const double a1 = double(K - k) / K, a2 = double(k) / K;
            for (int j = 0; j < 7; j++)
                x[j] = a1 * x0[j] + a2 * xf[j];
            u[0] = 0.;
            u[1] = 0.;
            u[2] = (T_max + T_min) / 2.;
        }
        else
        {
            // warm start: the stored trajectory is dimensional -> nondimensionalizeTrajectory
            x[0] /= m_scale;
            for (int j = 1; j < 7; j++)
                x[j] /= r_scale;
            for (int j = 0; j < 3; j++)
                u[j] /= m_scale * r_scale;
        }
        if (!so.interpolate_input && k == K - 1) // zero-order hold: the input slot of node K-1 is unused
            u[0] = u[1] = u[2] = 0.;
        double *uh = b.uhat + (i * K + k) * 3;
        if (mp.exact_minimum_thrust)
        {
            const double z = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
            const double s = z > 0. ? 1. / sqrt(z) : 1.;
            for (int j = 0; j < 3; j++)
                uh[j] = u[j] * s;
        }
        else
        {
            uh[0] = 0.;
            uh[1] = 0.;
            uh[2] = 1.;
        }
    }
    if (k0 == 0)
    {
        if (!warm)
        {
            b.sigma[i] = mp.final_time;
            b.wtrx[i] = so.weight_trust_region_trajectory;
        }
        b.active[i] = 1;
        b.converged[i] = 0;
        b.sc_iters[i] = 0;
        b.ipm_iters[i] = 0;
        b.status[i] = 0;
        b.norm1_nu[i] = 0.;
        b.sum_delta[i] = 0.;
        b.delta_sigma[i] = 0.;
    }
}
struct PersistLander3dof : ipm::Lander3dofSC
{
};
struct Lander3dofPlugin
{
    static constexpr int ID = SCPP_MODEL_LANDER3DOF;
    using Model = Lander3dofModel;
    using Table = ipm::Lander3dofSC;
    using PersistTable = PersistLander3dof;
    using Params = scpp_lander3dof_params;
    static constexpr int NX = Model::NX, NU = Model::NU;
    // SCVX_PERSISTENT: the persistent SCvx kernel is instantiated for this model (scvx_persistent.h; the default streaming engine).  false would run the
    // pool engine (scpp_hip_scvx_solve's rounds, the streaming engine's slot pools) -- same rows, bitwise
    static constexpr bool SPLIT_SCHEDULE = false, SC_PERSISTENT = false, SCVX_PERSISTENT = true;
    static bool supported(const Params &) { return true; }
    static __device__ void setupOne(const SCBuffers &b, const Params &mp, const scpp_sc_opts &so, int warm, long i, const double *xi, int k0, int kstep)
    {
        scSetupOneLander(b, mp, so, warm, i, xi, k0, kstep);
    }
    static __host__ __device__ double rx(int j, double ms, double rs) { return j == 0 ? ms : rs; }
    static __host__ __device__ double ru(int, double ms, double rs) { return ms * rs; }
};

// THE registry: the one place a model is named
template <class... PL>
struct PluginList
{
};
using Plugins = PluginList<RocketQuatPlugin, Rocket2dPlugin, Lander3dofPlugin>;

// batched set-up, one thread per instance: cold (warm = 0) or warm start of SCAlgorithm::solve / SCvxAlgorithm::solve
template <class PL>
__global__ void sc_setup_kernel(SCBuffers b, typename PL::Params mp, scpp_sc_opts so, int warm)
{
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= b.B)
        return;
    PL::setupOne(b, mp, so, warm, i, b.x_init_dim + i * PL::NX, 0, 1);
}

// redimensionalizeTrajectory, in place (entries whose factor is 1 are multiplied by 1: exact)
template <class PL>
__global__ void sc_redim_kernel(SCBuffers b)
{
    using namespace ipm;
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= b.B)
        return;
    const double m_scale = b.ip[i * IP_N + IP_MSCALE], r_scale = b.ip[i * IP_N + IP_RSCALE];
    for (int k = 0; k < b.K; k++)
    {
        double *x = b.X + (i * b.K + k) * PL::NX, *u = b.U + (i * b.K + k) * PL::NU;
        for (int j = 0; j < PL::NX; j++)
            x[j] *= PL::rx(j, m_scale, r_scale);
        for (int j = 0; j < PL::NU; j++)
            u[j] *= PL::ru(j, m_scale, r_scale);
    }
}

// counts active instances (one block)
__global__ void count_active_kernel(int B, const int *active, int *out)
{
    int acc = 0;
    for (int i = threadIdx.x; i < B; i += blockDim.x)
        acc += active[i] != 0;
    acc = int(wave_sum(double(acc)) + 0.5);
    __shared__ int part[16];
    const int wave = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0)
        part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int tot = 0;
        for (int w = 0; w < int(blockDim.x) / WAVE; w++)
            tot += part[w];
        *out = tot;
    }
}

} // namespace scpp
