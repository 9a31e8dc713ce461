// Host half of the Rocket2d model plugin (configuration, operating point); the flow map is device code
// (scpp_amd/csrc/model_rocketquat.h: Rocket2dModel).  Names a driver touches in the reference:
//   loadParameters, p.x_init, p.x_final, p.constrain_initial_final   scpp_models/src/rocket2d.cpp:138-198
//   getNewModelParameters (par = m, J_B, g_I, r_T_B)                 scpp_models/src/rocket2d.cpp:143-148
//   getOperatingPoint                                                scpp_models/src/rocket2d.cpp:40-44
#pragma once
#include <array>
#include <cmath>
#include <memory>
#include <string>

#include "parameter_server.hpp"
#include "rocket_quat.hpp" // counterUniform
#include "scpp_hip.h"

namespace scpp
{
namespace models
{

class Rocket2d
{
public:
    static constexpr int state_dim = 6, input_dim = 2, param_dim = 6;
    static constexpr int model_id = SCPP_MODEL_ROCKET2D;
    static constexpr bool has_scvx = true; // scpp_models/config/Rocket2D/SCvx.info
    using state_vector_t = std::array<double, 6>;
    using input_vector_t = std::array<double, 2>;
    using param_vector_t = std::array<double, 6>;
    using ptr_t = std::shared_ptr<Rocket2d>;

    struct Parameters
    {
        double m = 0., J_B = 0., g_I[2] = {0., 0.}, r_T_B[2] = {0., 0.};
        double T_min = 0., T_max = 0., gamma_gs = 0., tan_gamma_gs = 0., gimbal_max = 0., theta_max = 0., w_B_max = 0.;
        double final_time = 0.;
        state_vector_t x_init{}, x_final{};
        bool constrain_initial_final = true, add_slack_variables = false;
        // synthetic neighbours of the shipped x_init (the reference has no recipe for this model)
        void randomizeInitialState(uint64_t seed, uint64_t instance)
        {
            x_init[0] *= counterUniform(seed, instance, 0);
            x_init[2] = 0.05 * std::fabs(x_init[3]) * counterUniform(seed, instance, 1);
            x_init[3] *= 1. + 0.2 * counterUniform(seed, instance, 2);
            x_init[4] *= counterUniform(seed, instance, 3);
        }
    } p;

    // what scpp_hip_sc_setup_rocket2d consumes (SI units, radians)
    scpp_rocket2d_params abi() const
    {
        scpp_rocket2d_params a{};
        for (int j = 0; j < 2; j++)
        {
            a.g_I[j] = p.g_I[j];
            a.r_T_B[j] = p.r_T_B[j];
        }
        a.m = p.m;
        a.J_B = p.J_B;
        a.T_min = p.T_min;
        a.T_max = p.T_max;
        a.gimbal_max = p.gimbal_max;
        a.theta_max = p.theta_max;
        a.gamma_gs = p.gamma_gs;
        a.w_B_max = p.w_B_max;
        for (int j = 0; j < 6; j++)
            a.x_final[j] = p.x_final[size_t(j)];
        a.final_time = p.final_time;
        return a;
    }
    int scSetup(scpp_hip_ctx *ctx, const scpp_sc_opts *opts, const double *x_init, int B, int warm) const
    {
        if (!p.constrain_initial_final)
            return SCPP_E_UNSUPPORTED; // model.info:55-56: "enable for SC and disable for MPC/LQR"
        const scpp_rocket2d_params a = abi();
        return scpp_hip_sc_setup_rocket2d(ctx, &a, opts, x_init, B, warm);
    }
    // Parameters::nondimensionalize (rocket2d.cpp:200-203) and redimensionalizeTrajectory (:108-118)
    static void scales(const state_vector_t &x, const Rocket2d &m, double &m_scale, double &r_scale)
    {
        m_scale = m.p.m;
        r_scale = std::sqrt(x[0] * x[0] + x[1] * x[1]);
    }
    static void redimensionalize(state_vector_t &x, input_vector_t &u, double m_scale, double r_scale)
    {
        for (int j = 0; j < 4; j++)
            x[size_t(j)] *= r_scale;
        u[1] *= m_scale * r_scale;
    }
    void flowParams(double *par) const
    {
        param_vector_t q;
        getNewModelParameters(q);
        for (int j = 0; j < 6; j++)
            par[j] = q[size_t(j)];
    }

    static std::string getModelName() { return "Rocket2D"; }
    static std::string &parameterFolder()
    {
        static std::string folder = "../scpp_amd/config";
        return folder;
    }
    static void setParameterFolder(const std::string &f) { parameterFolder() = f; }
    static std::string getParameterFolder() { return parameterFolder() + "/" + getModelName(); }

    void loadParameters()
    {
        ParameterServer ps(getParameterFolder() + "/model.info");
        const double d2r = M_PI / 180.;
        double r_init[2], v_init[2], r_final[2], v_final[2], eta_init, eta_final, w_init, w_final;
        ps.loadMatrix("g_I", p.g_I, 2);
        ps.loadScalar("J_B", p.J_B);
        ps.loadMatrix("r_T_B", p.r_T_B, 2);
        ps.loadMatrix("r_init", r_init, 2);
        ps.loadMatrix("v_init", v_init, 2);
        ps.loadScalar("eta_init", eta_init);
        ps.loadScalar("w_init", w_init);
        ps.loadMatrix("r_final", r_final, 2);
        ps.loadMatrix("v_final", v_final, 2);
        ps.loadScalar("eta_final", eta_final);
        ps.loadScalar("w_final", w_final);
        ps.loadScalar("final_time", p.final_time);
        ps.loadScalar("m", p.m);
        ps.loadScalar("T_min", p.T_min);
        ps.loadScalar("T_max", p.T_max);
        ps.loadScalar("gamma_gs", p.gamma_gs);
        ps.loadScalar("gimbal_max", p.gimbal_max);
        ps.loadScalar("theta_max", p.theta_max);
        ps.loadScalar("w_B_max", p.w_B_max);
        ps.loadScalar("constrain_initial_final", p.constrain_initial_final);
        ps.loadScalar("add_slack_variables", p.add_slack_variables);
        p.gimbal_max *= d2r;
        p.theta_max *= d2r;
        p.gamma_gs *= d2r;
        p.w_B_max *= d2r;
        p.tan_gamma_gs = std::tan(p.gamma_gs);
        p.x_init = {r_init[0], r_init[1], v_init[0], v_init[1], eta_init * d2r, w_init * d2r};
        p.x_final = {r_final[0], r_final[1], v_final[0], v_final[1], eta_final * d2r, w_final * d2r};
    }

    void getNewModelParameters(param_vector_t &par) const { par = {p.m, p.J_B, p.g_I[0], p.g_I[1], p.r_T_B[0], p.r_T_B[1]}; }

    // The reference streams `0, -p.g_I * p.m` (a scalar and a 2-vector) into a 2-vector, which asserts in debug builds and
    // overruns otherwise; the evident intent, the hover input, is returned.
    void getOperatingPoint(state_vector_t &x, input_vector_t &u) const
    {
        x.fill(0.);
        u = {0., -p.g_I[1] * p.m};
    }

    // synthetic start states around the shipped x_init (same recipe as scpp_amd/models.py: Rocket2D.randomized_initial_states)
    state_vector_t randomizedInitialState(uint64_t seed, uint64_t instance) const
    {
        state_vector_t x = p.x_init;
        x[0] *= counterUniform(seed, instance, 0);
        x[2] = 0.05 * std::fabs(x[3]) * counterUniform(seed, instance, 1);
        x[3] *= 1. + 0.2 * counterUniform(seed, instance, 2);
        x[4] *= counterUniform(seed, instance, 3);
        return x;
    }
};

} // namespace models
} // namespace scpp
