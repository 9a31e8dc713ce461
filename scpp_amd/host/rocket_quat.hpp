// Host half of the RocketQuat model plugin (configuration, initial states); the flow map is device code
// (scpp_amd/csrc/model_rocketquat.h).  Mirrors the names a driver touches in the reference:
//   Model::getModelName / setParameterFolder / getParameterFolder   scpp_core/include/systemModel.hpp:142-158
//   loadParameters, p.x_init, p.x_final                              scpp_models/src/rocketQuat.cpp:229-289
//   Parameters::randomizeInitialState                                scpp_models/src/rocketQuat.cpp:203-227
//   eulerToQuaternionXYZ                                             scpp_models/include/common.hpp:30-38
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>

#include "parameter_server.hpp"
#include "scpp_hip.h"

namespace scpp
{
namespace models
{

inline std::array<double, 4> eulerToQuaternionXYZ(const double eta[3])
{
    const double cx = std::cos(0.5 * eta[0]), sx = std::sin(0.5 * eta[0]);
    const double cy = std::cos(0.5 * eta[1]), sy = std::sin(0.5 * eta[1]);
    const double cz = std::cos(0.5 * eta[2]), sz = std::sin(0.5 * eta[2]);
    const double aw = cx * cy, ax = sx * cy, ay = cx * sy, az = sx * sy;
    return {aw * cz - az * sz, ax * cz + ay * sz, ay * cz - ax * sz, aw * sz + az * cz};
}

// counter-based uniform in [-1,1): SplitMix64 keyed by (seed, instance, draw)   
inline double counterUniform(uint64_t seed, uint64_t instance, uint64_t draw)
{
    uint64_t z = seed + (instance * 8 + draw + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return 2. * (double(z >> 11) * (1. / 9007199254740992.)) - 1.;
}

class RocketQuat
{
public:
    static constexpr int state_dim = 14, input_dim = 4, param_dim = 10;
    static constexpr int model_id = SCPP_MODEL_ROCKETQUAT;
    static constexpr bool has_scvx = true; // ships an SCvx.info
    using state_vector_t = std::array<double, 14>;
    using input_vector_t = std::array<double, 4>;
    using ptr_t = std::shared_ptr<RocketQuat>;

    struct Parameters
    {
        scpp_rocketquat_params abi{};   // what the device consumes (SI units, radians)
        state_vector_t x_init{}, x_final{};
        double rpy_init[3] = {0., 0., 0.};
        bool random_initial_state = false;

        // the reference's recipe with a reproducible counter-based generator
        void randomizeInitialState(uint64_t seed, uint64_t instance)
        {
            x_init[1] *= counterUniform(seed, instance, 0);
            x_init[2] *= counterUniform(seed, instance, 1);
            x_init[4] *= counterUniform(seed, instance, 2);
            x_init[5] *= counterUniform(seed, instance, 3);
            x_init[6] *= 1. + 0.2 * counterUniform(seed, instance, 4);
            const double euler[3] = {counterUniform(seed, instance, 5) * rpy_init[0], counterUniform(seed, instance, 6) * rpy_init[1], rpy_init[2]};
            const auto q = eulerToQuaternionXYZ(euler);
            for (int j = 0; j < 4; j++)
                x_init[7 + j] = q[j];
        }
    } p;

    static std::string getModelName() { return "RocketQuat"; }
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
        scpp_rocketquat_params &a = p.abi;
        double r_init[3], v_init[3], rpy_init[3], w_init[3], w_final[3], r_final[3], v_final[3], rpy_final[3];
        double m_init, m_dry, I_sp;
        bool exact_min, roll;
        ps.loadMatrix("g_I", a.g_I, 3);
        ps.loadMatrix("J_B", a.J_B, 3);
        ps.loadMatrix("r_T_B", a.r_T_B, 3);
        ps.loadScalar("m_init", m_init);
        ps.loadMatrix("r_init", r_init, 3);
        ps.loadMatrix("v_init", v_init, 3);
        ps.loadMatrix("rpy_init", rpy_init, 3);
        ps.loadMatrix("w_init", w_init, 3);
        ps.loadMatrix("w_final", w_final, 3);
        ps.loadScalar("m_dry", m_dry);
        ps.loadMatrix("r_final", r_final, 3);
        ps.loadMatrix("v_final", v_final, 3);
        ps.loadMatrix("rpy_final", rpy_final, 3);
        ps.loadScalar("T_min", a.T_min);
        ps.loadScalar("T_max", a.T_max);
        ps.loadScalar("t_max", a.t_max);
        ps.loadScalar("I_sp", I_sp);
        ps.loadScalar("gimbal_max", a.gimbal_max);
        ps.loadScalar("theta_max", a.theta_max);
        ps.loadScalar("gamma_gs", a.gamma_gs);
        ps.loadScalar("w_B_max", a.w_B_max);
        ps.loadScalar("random_initial_state", p.random_initial_state);
        ps.loadScalar("final_time", a.final_time);
        ps.loadScalar("exact_minimum_thrust", exact_min);
        ps.loadScalar("enable_roll_control", roll);
        a.exact_minimum_thrust = exact_min;
        a.enable_roll_control = roll;
        a.gimbal_max *= d2r;
        a.theta_max *= d2r;
        a.gamma_gs *= d2r;
        a.w_B_max *= d2r;
        for (int j = 0; j < 3; j++)
        {
            rpy_init[j] *= d2r;
            rpy_final[j] *= d2r;
            w_init[j] *= d2r;
            w_final[j] *= d2r;
            p.rpy_init[j] = rpy_init[j];
        }
        a.alpha_m = 1. / (I_sp * std::fabs(a.g_I[2]));
        const auto q0 = eulerToQuaternionXYZ(rpy_init), q1 = eulerToQuaternionXYZ(rpy_final);
        p.x_init[0] = m_init;
        p.x_final[0] = m_dry;
        for (int j = 0; j < 3; j++)
        {
            p.x_init[1 + j] = r_init[j];
            p.x_init[4 + j] = v_init[j];
            p.x_init[11 + j] = w_init[j];
            p.x_final[1 + j] = r_final[j];
            p.x_final[4 + j] = v_final[j];
            p.x_final[11 + j] = w_final[j];
        }
        for (int j = 0; j < 4; j++)
        {
            p.x_init[7 + j] = q0[j];
            p.x_final[7 + j] = q1[j];
        }
        for (int j = 0; j < 14; j++)
            a.x_final[j] = p.x_final[j];
    }

    // the C-ABI set-up call of this model (scpp_hip_sc_setup) and the scales of Parameters::nondimensionalize (:291-294)
    int scSetup(scpp_hip_ctx *ctx, const scpp_sc_opts *opts, const double *x_init, int B, int warm) const
    {
        return scpp_hip_sc_setup(ctx, &p.abi, opts, x_init, B, warm);
    }
    static void scales(const state_vector_t &x, const RocketQuat &, double &m_scale, double &r_scale)
    {
        m_scale = x[0];
        r_scale = std::sqrt(x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    }
    // redimensionalizeTrajectory (rocketQuat.cpp:188-201) of one node
    static void redimensionalize(state_vector_t &x, input_vector_t &u, double m_scale, double r_scale)
    {
        x[0] *= m_scale;
        for (int j = 1; j < 7; j++)
            x[size_t(j)] *= r_scale;
        for (int j = 0; j < 3; j++)
            u[size_t(j)] *= m_scale * r_scale;
        u[3] *= m_scale * r_scale * r_scale;
    }

    // flow-map parameters in SI units for the plant simulation (rocketQuat.cpp:168-173 without scaling)
    void flowParams(double *par) const
    {
        par[0] = p.abi.alpha_m;
        for (int j = 0; j < 3; j++)
        {
            par[1 + j] = p.abi.g_I[j];
            par[4 + j] = p.abi.J_B[j];
            par[7 + j] = p.abi.r_T_B[j];
        }
    }
};

} // namespace models
} // namespace scpp
