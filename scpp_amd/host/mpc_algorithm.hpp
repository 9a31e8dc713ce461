// scpp::MPCAlgorithm with the reference's member names (scpp_core/include/MPCAlgorithm.hpp:17-64,
// src/MPCAlgorithm.cpp:11-139) over the C ABI, plus batched overloads: every call serves B independent controllers.
// All arithmetic runs in libscpp_hip.so (scpp_hip_mpc_*); a missing library is a link error, there is no CPU path.
#pragma once
#include <stdexcept>
#include <vector>

#include "rocket_2d.hpp"

namespace scpp
{

class MPCAlgorithm
{
public:
    using Model = models::Rocket2d;
    using state_vector_t = Model::state_vector_t;
    using input_vector_t = Model::input_vector_t;
    using state_vector_v_t = std::vector<state_vector_t>;
    using input_vector_v_t = std::vector<input_vector_t>;

    explicit MPCAlgorithm(Model::ptr_t model_, int batch_max_ = 1, int device_ = 0) : model(model_), batch_max(batch_max_), device(device_)
    {
        loadParameters();
    }
    ~MPCAlgorithm()
    {
        if (ctx)
            scpp_hip_destroy(ctx);
    }
    MPCAlgorithm(const MPCAlgorithm &) = delete;
    MPCAlgorithm &operator=(const MPCAlgorithm &) = delete;

    // MPCAlgorithm.cpp:34-69
    void initialize()
    {
        if (!(state_weights_set && input_weights_set))
            throw std::runtime_error("MPCAlgorithm: weights not set");
        if (model->p.constrain_initial_final)
            throw std::runtime_error("MPCAlgorithm: constrain_initial_final must be disabled for MPC (model.info: 'enable for SC and "
                                     "disable for MPC/LQR')");
        scpp_mpc_opts o{};
        o.K = int32_t(K);
        o.nondimensionalize = nondimensionalize;
        o.constant_dynamics = constant_dynamics;
        o.intermediate_cost_active = intermediate_cost_active;
        o.time_horizon = time_horizon;
        for (int i = 0; i < 6; i++)
        {
            o.state_weights_intermediate[i] = state_weights_intermediate[size_t(i)];
            o.state_weights_terminal[i] = state_weights_terminal[size_t(i)];
        }
        o.input_weights[0] = input_weights[0];
        o.input_weights[1] = input_weights[1];
        state_vector_t x_eq;
        input_vector_t u_eq;
        model->getOperatingPoint(x_eq, u_eq);
        for (int i = 0; i < 6; i++)
            o.x_eq[i] = x_eq[size_t(i)];
        o.u_eq[0] = u_eq[0];
        o.u_eq[1] = u_eq[1];
        const auto &p = model->p;
        o.tan_gamma_gs = p.tan_gamma_gs;
        o.theta_max = p.theta_max;
        o.w_B_max = p.w_B_max;
        o.gimbal_max = p.gimbal_max;
        o.T_min = p.T_min;
        o.T_max = p.T_max;
        o.x_scale_ref = std::hypot(p.x_init[0], p.x_init[1]);
        Model::param_vector_t par;
        model->getNewModelParameters(par);
        check(scpp_hip_create(&ctx, device, SCPP_MODEL_ROCKET2D, int(K), batch_max, 0), "scpp_hip_create");
        check(scpp_hip_mpc_setup(ctx, &o, par.data()), "scpp_hip_mpc_setup");
        check(scpp_hip_mpc_get_model(ctx, A, B, z), "scpp_hip_mpc_get_model");
        initialized = true;
    }

    void setInitialState(const state_vector_t &x) { x_init.assign(1, x); }
    void setInitialState(const state_vector_v_t &x) { x_init = x; } // batched
    void setFinalState(const state_vector_t &x) { x_final = x; }
    void setStateWeights(const state_vector_t &intermediate, const state_vector_t &terminal)
    {
        state_weights_intermediate = intermediate;
        state_weights_terminal = terminal;
        state_weights_set = true;
    }
    void setInputWeights(const input_vector_t &intermediate)
    {
        input_weights = intermediate;
        input_weights_set = true;
    }

    // MPCAlgorithm.cpp:95-121 for every initial state; returns the number of successful solves
    int solve()
    {
        if (!initialized)
            throw std::runtime_error("MPCAlgorithm: not initialized");
        const size_t Bn = x_init.size();
        std::vector<double> xi(Bn * 6), xf(Bn * 6);
        for (size_t b = 0; b < Bn; b++)
            for (size_t i = 0; i < 6; i++)
            {
                xi[b * 6 + i] = x_init[b][i];
                xf[b * 6 + i] = x_final[i];
            }
        int n = 0;
        check(scpp_hip_mpc_solve(ctx, xi.data(), xf.data(), int(Bn), &n), "scpp_hip_mpc_solve");
        Xs.assign(Bn * K * 6, 0.);
        Us.assign(Bn * (K - 1) * 2, 0.);
        status.assign(Bn, 0);
        iterations.assign(Bn, 0);
        check(scpp_hip_mpc_download(ctx, Xs.data(), Us.data(), nullptr, status.data(), iterations.data()), "scpp_hip_mpc_download");
        return n;
    }

    // MPCAlgorithm.cpp:134-138 (controller `b` of the batch)
    void getSolution(state_vector_v_t &X, input_vector_v_t &U, size_t b = 0) const
    {
        X.resize(K);
        U.resize(K - 1);
        for (size_t k = 0; k < K; k++)
            for (size_t i = 0; i < 6; i++)
                X[k][i] = Xs[(b * K + k) * 6 + i];
        for (size_t k = 0; k + 1 < K; k++)
            for (size_t i = 0; i < 2; i++)
                U[k][i] = Us[(b * (K - 1) + k) * 2 + i];
    }

    // scpp::simulate for the batch (MPC_sim.cpp:67)
    void simulateBatch(double dt, const input_vector_v_t &u0, const input_vector_v_t &u1, state_vector_v_t &x)
    {
        const size_t Bn = x.size();
        std::vector<double> dtv(Bn, dt), a(Bn * 2), c(Bn * 2), xs(Bn * 6);
        for (size_t b = 0; b < Bn; b++)
        {
            for (size_t i = 0; i < 2; i++)
            {
                a[b * 2 + i] = u0[b][i];
                c[b * 2 + i] = u1[b][i];
            }
            for (size_t i = 0; i < 6; i++)
                xs[b * 6 + i] = x[b][i];
        }
        check(scpp_hip_simulate(ctx, dtv.data(), a.data(), c.data(), xs.data(), int(Bn)), "scpp_hip_simulate");
        for (size_t b = 0; b < Bn; b++)
            for (size_t i = 0; i < 6; i++)
                x[b][i] = xs[b * 6 + i];
    }

    Model::ptr_t model;
    size_t K = 0;
    bool nondimensionalize = false, constant_dynamics = true, intermediate_cost_active = false;
    double time_horizon = 0.;
    double A[36], B[12], z[6];
    std::vector<int32_t> status, iterations;
    scpp_hip_ctx *ctx = nullptr;

private:
    // MPCAlgorithm.cpp:17-32
    void loadParameters()
    {
        ParameterServer param(model->getParameterFolder() + "/MPC.info");
        param.loadScalar("K", K);
        param.loadScalar("nondimensionalize", nondimensionalize);
        param.loadScalar("constant_dynamics", constant_dynamics);
        param.loadScalar("intermediate_cost_active", intermediate_cost_active);
        param.loadScalar("time_horizon", time_horizon);
        state_vector_t wi, wt;
        input_vector_t wu;
        param.loadMatrix("state_weights_intermediate", wi.data(), 6);
        param.loadMatrix("state_weights_terminal", wt.data(), 6);
        param.loadMatrix("input_weights", wu.data(), 2);
        setStateWeights(wi, wt);
        setInputWeights(wu);
    }
    static void check(int rc, const char *what)
    {
        if (rc != SCPP_OK)
            throw std::runtime_error(std::string(what) + " failed with code " + std::to_string(rc));
    }
    int batch_max, device;
    bool initialized = false, state_weights_set = false, input_weights_set = false;
    state_vector_t state_weights_intermediate{}, state_weights_terminal{}, x_final{};
    input_vector_t input_weights{};
    state_vector_v_t x_init;
    std::vector<double> Xs, Us;
};

} // namespace scpp
