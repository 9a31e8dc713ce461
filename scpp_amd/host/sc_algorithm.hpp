// C++17 host front end over the C ABI (include/scpp_hip.h): the reference's SCAlgorithm interface
//   SCAlgorithm(Model::ptr_t), initialize(), solve(bool warm_start = false), getSolution(td&), getAllSolutions(v&)
//   (scpp_core/include/SCAlgorithm.hpp:17-45, scpp_core/src/SCAlgorithm.cpp:14-232)
// plus the batched overloads the device engine exists for.  All numerics run in libscpp_hip.so;
// there is no CPU fallback -- construction throws if the library cannot create a device context.
#pragma once
#include <array>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "parameter_server.hpp"
#include "rocket_2d.hpp"
#include "rocket_quat.hpp"
#include "scpp_hip.h"

// The reference selects the active model with a compile definition (CMakeLists.txt:33-55, scpp_core/include/activeModel.hpp:
// its default is Rocket2d).  Same here: -DSCPP_ACTIVE_MODEL_ROCKET2D builds the front end for Rocket2d, otherwise RocketQuat.
// On the device both run the same solver, instantiated for the model's constraint table (csrc/constraint_table.h).
#ifdef SCPP_ACTIVE_MODEL_ROCKET2D
using Model = scpp::models::Rocket2d;
#else
using Model = scpp::models::RocketQuat;
#endif

// trajectoryData.hpp:8-32
struct trajectory_data_t
{
    std::vector<Model::state_vector_t> X;
    std::vector<Model::input_vector_t> U;
    double t = 0.;
    void initialize(size_t K, bool interpolate_input)
    {
        X.assign(K, Model::state_vector_t{});
        U.assign(interpolate_input ? K : K - 1, Model::input_vector_t{});
        t = 0.;
    }
    bool interpolatedInput() const { return U.size() == X.size(); }
    size_t n_X() const { return X.size(); }
    size_t n_U() const { return U.size(); }
};

namespace scpp
{

struct batch_result_t
{
    std::vector<trajectory_data_t> td;
    std::vector<int32_t> sc_iterations, converged, status, ipm_iterations;
    std::vector<double> norm1_nu, sum_delta;
};

class SCAlgorithm
{
public:
    explicit SCAlgorithm(Model::ptr_t model_, int batch_max_ = 1, int device_ = 0, int K_override_ = 0)
        : model(std::move(model_)), batch_max(batch_max_), device(device_), K_override(K_override_) {}
    ~SCAlgorithm()
    {
        if (ctx)
            scpp_hip_destroy(ctx);
    }
    SCAlgorithm(const SCAlgorithm &) = delete;
    SCAlgorithm &operator=(const SCAlgorithm &) = delete;

    // SCAlgorithm.cpp:22-64
    void loadParameters()
    {
        ParameterServer ps(Model::getParameterFolder() + "/SC.info");
        bool fft, ii, nd;
        ps.loadScalar("free_final_time", fft);
        ps.loadScalar("interpolate_input", ii);
        ps.loadScalar("K", opts.K);
        ps.loadScalar("nondimensionalize", nd);
        ps.loadScalar("weight_time", opts.weight_time);
        opts.weight_trust_region_time = 0.;
        if (fft) // SCAlgorithm.cpp:42-45: only read (and only present) for a free final time
            ps.loadScalar("weight_trust_region_time", opts.weight_trust_region_time);
        ps.loadScalar("weight_trust_region_trajectory", opts.weight_trust_region_trajectory);
        ps.loadScalar("weight_virtual_control", opts.weight_virtual_control);
        ps.loadScalar("nu_tol", opts.nu_tol);
        ps.loadScalar("delta_tol", opts.delta_tol);
        ps.loadScalar("max_iterations", opts.max_iterations);
        opts.free_final_time = fft;
        opts.interpolate_input = ii;
        opts.nondimensionalize = nd;
        if (K_override > 0)
            opts.K = K_override;
    }
    void initialize()
    {
        loadParameters();
        check(scpp_hip_create(&ctx, device, Model::model_id, opts.K, batch_max, 0), "scpp_hip_create");
        td.initialize(size_t(opts.K), opts.interpolate_input != 0);
    }
    // RKF78 steps per shooting segment: 5 = the reference's fixed count (discretizationImplementation.hpp:141,154; the default), 0 = the
    // engine's opt-in step-length rule (include/scpp_hip.h).  Call after initialize().
    void setDiscretizationSteps(int steps)
    {
        if (scpp_hip_set_discretization_steps(ctx, steps) != SCPP_OK)
            throw std::invalid_argument("setDiscretizationSteps: 0 (adaptive) or 1 .. 5");
    }
    // How the device schedules the loop (include/scpp_hip.h): SCPP_STREAM_PERSISTENT (default: one kernel launch, a wavefront takes its instance through
    // the whole algorithm) or SCPP_STREAM_POOLS (rounds of launches); the results are bitwise the same.  Call after initialize().
    void setStreamEngine(int engine)
    {
        if (scpp_hip_set_stream_engine(ctx, engine) != SCPP_OK)
            throw std::invalid_argument("setStreamEngine: SCPP_STREAM_POOLS or SCPP_STREAM_PERSISTENT");
    }

    // ---- the reference's single-problem interface (instance = model->p.x_init) ----
    void solve(bool warm_start = false)
    {
        const std::vector<Model::state_vector_t> x{model->p.x_init};
        all_td.clear();
        setup(x, warm_start, nullptr);
        // drive the iterations from the host so that every iterate can be recorded (all_td, SCAlgorithm.cpp:160-170)
        all_td.push_back(downloadOne(true));
        int n_active = 1;
        iterations = 0;
        while (iterations < opts.max_iterations && n_active > 0)
        {
            iterations++;
            check(scpp_hip_sc_iterate(ctx, &n_active), "scpp_hip_sc_iterate");
            all_td.push_back(downloadOne(true));
        }
        int nconv = 0;
        check(scpp_hip_sc_finish(ctx, &nconv), "scpp_hip_sc_finish");
        converged = nconv > 0;
        td = downloadOne(false);
    }
    void getSolution(trajectory_data_t &trajectory) const { trajectory = td; }
    void getAllSolutions(std::vector<trajectory_data_t> &all_trajectories) { all_trajectories = all_td; }
    bool hasConverged() const { return converged; }
    int getIterations() const { return iterations; }

    // ---- batched interface: B independent instances that differ in x_init ----
    void solveBatch(const std::vector<Model::state_vector_t> &x_inits, batch_result_t &out, bool warm_start = false,
                    const std::vector<int32_t> *active = nullptr)
    {
        setup(x_inits, warm_start, active);
        int nconv = 0;
        check(scpp_hip_sc_solve(ctx, &nconv), "scpp_hip_sc_solve");
        download(int(x_inits.size()), out);
    }
    // scpp::simulate (scpp_core/src/simulation.cpp:25-42) for every row, SI units
    void simulateBatch(double dt, const std::vector<Model::input_vector_t> &u0, const std::vector<Model::input_vector_t> &u1,
                       std::vector<Model::state_vector_t> &x)
    {
        const int B = int(x.size());
        std::vector<double> par(size_t(B) * Model::param_dim), dts(size_t(B), dt);
        for (int b = 0; b < B; b++)
            model->flowParams(&par[size_t(b) * Model::param_dim]);
        check(scpp_hip_set_flow_params(ctx, par.data(), B), "scpp_hip_set_flow_params");
        check(scpp_hip_simulate(ctx, dts.data(), &u0[0][0], &u1[0][0], &x[0][0], B), "scpp_hip_simulate");
    }

    scpp_sc_opts opts{};
    Model::ptr_t model;

private:
    static void check(int rc, const char *what)
    {
        if (rc != SCPP_OK)
            throw std::runtime_error(std::string(what) + " failed with code " + std::to_string(rc));
    }
    void setup(const std::vector<Model::state_vector_t> &x, bool warm_start, const std::vector<int32_t> *active)
    {
        if (!ctx)
            throw std::runtime_error("SCAlgorithm::initialize() has not been called");
        const int B = int(x.size());
        check(model->scSetup(ctx, &opts, &x[0][0], B, warm_start ? 1 : 0), "scpp_hip_sc_setup");
        if (active)
            check(scpp_hip_sc_set_active(ctx, active->data(), B), "scpp_hip_sc_set_active");
        Model::scales(x[0], *model, scale_m, scale_r);
    }
    void download(int B, batch_result_t &out)
    {
        const size_t K = size_t(opts.K);
        const size_t nB = size_t(B);
        constexpr size_t NX = Model::state_dim, NU = Model::input_dim;
        std::vector<double> X(nB * K * NX), U(nB * K * NU), sigma(nB, 0.);
        out.sc_iterations.assign(size_t(B), 0);
        out.converged.assign(size_t(B), 0);
        out.status.assign(size_t(B), 0);
        out.ipm_iterations.assign(size_t(B), 0);
        out.norm1_nu.assign(size_t(B), 0.);
        out.sum_delta.assign(size_t(B), 0.);
        check(scpp_hip_download(ctx, X.data(), U.data(), sigma.data(), out.sc_iterations.data(), out.norm1_nu.data(),
                                out.converged.data(), out.status.data(), out.ipm_iterations.data(), out.sum_delta.data()),
              "scpp_hip_download");
        out.td.resize(size_t(B));
        for (int b = 0; b < B; b++)
        {
            trajectory_data_t &t = out.td[size_t(b)];
            t.initialize(K, opts.interpolate_input != 0); // zero-order hold: K - 1 inputs (trajectoryData.hpp:27-32); device slot K-1 unused
            for (size_t k = 0; k < K; k++)
            {
                for (size_t j = 0; j < NX; j++)
                    t.X[k][j] = X[(size_t(b) * K + k) * NX + j];
                if (k < t.U.size())
                    for (size_t j = 0; j < NU; j++)
                        t.U[k][j] = U[(size_t(b) * K + k) * NU + j];
            }
            t.t = sigma[size_t(b)];
        }
    }
    // instance 0; redimensionalizeTrajectory on the host for iterates fetched mid-solve (rocketQuat.cpp:322-332)
    trajectory_data_t downloadOne(bool redimensionalize)
    {
        batch_result_t r;
        download(1, r);
        trajectory_data_t t = r.td[0];
        if (redimensionalize && opts.nondimensionalize)
            for (size_t k = 0; k < t.X.size(); k++)
            {
                Model::input_vector_t none{}; // zero-order hold: no input at the last node
                Model::redimensionalize(t.X[k], k < t.U.size() ? t.U[k] : none, scale_m, scale_r);
            }
        return t;
    }

    int batch_max, device, K_override;
    scpp_hip_ctx *ctx = nullptr;
    trajectory_data_t td;
    std::vector<trajectory_data_t> all_td;
    bool converged = false;
    int iterations = 0;
    double scale_m = 1., scale_r = 1.;
};

// SCvxAlgorithm (scpp_core/include/SCvxAlgorithm.hpp:18-48, src/SCvxAlgorithm.cpp:14-260): same shape as SCAlgorithm;
// fixed final time, hard input trust region with rho-ratio radius update.  The whole iteration runs on the device
// (scpp_hip_scvx_setup / scpp_hip_scvx_solve).
struct scvx_result_t : batch_result_t
{
    std::vector<double> trust_region, nonlinear_cost;
    std::vector<int32_t> solves;
};

class SCvxAlgorithm
{
public:
    explicit SCvxAlgorithm(Model::ptr_t model_, int batch_max_ = 1, int device_ = 0, int K_override_ = 0)
        : model(std::move(model_)), batch_max(batch_max_), device(device_), K_override(K_override_) {}
    ~SCvxAlgorithm()
    {
        if (ctx)
            scpp_hip_destroy(ctx);
    }
    SCvxAlgorithm(const SCvxAlgorithm &) = delete;
    SCvxAlgorithm &operator=(const SCvxAlgorithm &) = delete;

    // SCvxAlgorithm.cpp:22-44
    void loadParameters()
    {
        ParameterServer ps(Model::getParameterFolder() + "/SCvx.info");
        bool ii, nd;
        ps.loadScalar("K", opts.K);
        ps.loadScalar("nondimensionalize", nd);
        ps.loadScalar("max_iterations", opts.max_iterations);
        ps.loadScalar("alpha", opts.alpha);
        ps.loadScalar("beta", opts.beta);
        ps.loadScalar("rho_0", opts.rho_0);
        ps.loadScalar("rho_1", opts.rho_1);
        ps.loadScalar("rho_2", opts.rho_2);
        ps.loadScalar("change_threshold", opts.change_threshold);
        ps.loadScalar("weight_virtual_control", opts.weight_virtual_control);
        ps.loadScalar("trust_region", opts.trust_region);
        ps.loadScalar("interpolate_input", ii);
        opts.interpolate_input = ii;
        opts.nondimensionalize = nd;
        if (K_override > 0)
            opts.K = K_override;
    }
    void initialize()
    {
        loadParameters();
        if (!Model::has_scvx)
            throw std::runtime_error("SCvxAlgorithm: the active model ships no SCvx configuration");
        const int rc = scpp_hip_create(&ctx, device, Model::model_id, opts.K, batch_max, 0);
        if (rc != SCPP_OK)
            throw std::runtime_error("scpp_hip_create failed with code " + std::to_string(rc));
    }
    // RKF78 steps per shooting segment: 5 = the reference's fixed count (discretizationImplementation.hpp:141,154; the default), 0 = the
    // engine's opt-in step-length rule (include/scpp_hip.h).  Call after initialize().
    void setDiscretizationSteps(int steps)
    {
        if (scpp_hip_set_discretization_steps(ctx, steps) != SCPP_OK)
            throw std::invalid_argument("setDiscretizationSteps: 0 (adaptive) or 1 .. 5");
    }
    // How the device schedules the loop (include/scpp_hip.h): SCPP_STREAM_PERSISTENT (default: one kernel launch, a wavefront takes its instance through
    // the whole algorithm) or SCPP_STREAM_POOLS (rounds of launches); the results are bitwise the same.  Call after initialize().
    void setStreamEngine(int engine)
    {
        if (scpp_hip_set_stream_engine(ctx, engine) != SCPP_OK)
            throw std::invalid_argument("setStreamEngine: SCPP_STREAM_POOLS or SCPP_STREAM_PERSISTENT");
    }
    void solve(bool warm_start = false)
    {
        scvx_result_t r;
        recordIterates(true); // the single-instance front end keeps all_td like the reference (SCvxAlgorithm.cpp:192,201)
        solveBatch({model->p.x_init}, r, warm_start);
        td = r.td[0];
        converged = r.converged[0] != 0;
        iterations = r.sc_iterations[0];
        std::vector<std::vector<trajectory_data_t>> all;
        getAllSolutionsBatch(all);
        all_td = all.at(0);
    }
    void getSolution(trajectory_data_t &trajectory) const { trajectory = td; }
    // SCvxAlgorithm::getAllSolutions (SCvxAlgorithm.hpp:48, SCvxAlgorithm.cpp:245-260): the initial trajectory and the trajectory after every iteration
    // (= after every accepted candidate), redimensionalised
    void getAllSolutions(std::vector<trajectory_data_t> &all_trajectories) { all_trajectories = all_td; }
    // the batched form: opt in BEFORE solveBatch (the device then records max_iterations + 1 trajectories per instance, include/scpp_hip.h) ...
    void recordIterates(bool enable)
    {
        if (!ctx)
            throw std::runtime_error("SCvxAlgorithm::initialize() has not been called");
        if (enable != recording && scpp_hip_scvx_record_iterates(ctx, enable ? 1 : 0) != SCPP_OK)
            throw std::runtime_error("scpp_hip_scvx_record_iterates failed");
        recording = enable;
    }
    // ... and read the record of the last solveBatch: all[b] = all_td of instance b
    void getAllSolutionsBatch(std::vector<std::vector<trajectory_data_t>> &all)
    {
        if (!recording || last_B < 1)
            throw std::runtime_error("getAllSolutionsBatch: recordIterates(true) before solveBatch");
        const size_t K = size_t(opts.K), nB = size_t(last_B), cap = size_t(opts.max_iterations) + 1;
        constexpr size_t NX = Model::state_dim, NU = Model::input_dim;
        std::vector<double> X(nB * cap * K * NX), U(nB * cap * K * NU), sigma(nB, 0.);
        std::vector<int32_t> n(nB, 0);
        int rc = scpp_hip_scvx_download_iterates(ctx, 0, last_B, int(cap), X.data(), U.data(), nullptr, n.data());
        if (rc == SCPP_OK)
            rc = scpp_hip_download(ctx, nullptr, nullptr, sigma.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc != SCPP_OK)
            throw std::runtime_error("scpp_hip_scvx_download_iterates failed with code " + std::to_string(rc));
        all.assign(nB, {});
        for (size_t b = 0; b < nB; b++)
            for (size_t j = 0; j < size_t(n[b]) && j < cap; j++)
            {
                trajectory_data_t t;
                t.initialize(K, opts.interpolate_input != 0);
                for (size_t k = 0; k < K; k++)
                    for (size_t e = 0; e < NX; e++)
                        t.X[k][e] = X[((b * cap + j) * K + k) * NX + e];
                for (size_t k = 0; k < t.U.size(); k++)
                    for (size_t e = 0; e < NU; e++)
                        t.U[k][e] = U[((b * cap + j) * K + k) * NU + e];
                t.t = sigma[b];
                all[b].push_back(t);
            }
    }
    bool hasConverged() const { return converged; }
    int getIterations() const { return iterations; }

    void solveBatch(const std::vector<Model::state_vector_t> &x_inits, scvx_result_t &out, bool warm_start = false)
    {
        if (!ctx)
            throw std::runtime_error("SCvxAlgorithm::initialize() has not been called");
        if (!warm_start)
            loadParameters(); // cold start re-reads SCvx.info (:179)
        const int B = int(x_inits.size());
        last_B = B;
        int rc = scvxSetup(*model, &x_inits[0][0], B, warm_start);
        int nconv = 0;
        if (rc == SCPP_OK)
            rc = scpp_hip_scvx_solve(ctx, &nconv);
        if (rc != SCPP_OK)
            throw std::runtime_error("scpp_hip_scvx solve failed with code " + std::to_string(rc));
        const size_t K = size_t(opts.K), nB = size_t(B);
        constexpr size_t NX = Model::state_dim, NU = Model::input_dim;
        std::vector<double> X(nB * K * NX), U(nB * K * NU), sigma(nB, 0.);
        out.sc_iterations.assign(nB, 0);
        out.converged.assign(nB, 0);
        out.status.assign(nB, 0);
        out.ipm_iterations.assign(nB, 0);
        out.norm1_nu.assign(nB, 0.);
        out.sum_delta.assign(nB, 0.);
        out.trust_region.assign(nB, 0.);
        out.nonlinear_cost.assign(nB, 0.);
        out.solves.assign(nB, 0);
        rc = scpp_hip_download(ctx, X.data(), U.data(), sigma.data(), out.sc_iterations.data(), out.norm1_nu.data(),
                               out.converged.data(), out.status.data(), out.ipm_iterations.data(), out.sum_delta.data());
        if (rc == SCPP_OK)
            rc = scpp_hip_scvx_download_state(ctx, out.trust_region.data(), out.nonlinear_cost.data(), out.solves.data(), nullptr);
        if (rc != SCPP_OK)
            throw std::runtime_error("scpp_hip download failed with code " + std::to_string(rc));
        out.td.resize(nB);
        for (size_t b = 0; b < nB; b++)
        {
            trajectory_data_t &t = out.td[b];
            t.initialize(K, opts.interpolate_input != 0); // zero-order hold: K - 1 inputs (trajectoryData.hpp:27-32); the device's slot K-1 is unused
            for (size_t k = 0; k < K; k++)
                for (size_t j = 0; j < NX; j++)
                    t.X[k][j] = X[(b * K + k) * NX + j];
            for (size_t k = 0; k < t.U.size(); k++)
                for (size_t j = 0; j < NU; j++)
                    t.U[k][j] = U[(b * K + k) * NU + j];
            t.t = sigma[b];
        }
    }

    scpp_scvx_opts opts{};
    Model::ptr_t model;

private:
    int scvxSetup(const scpp::models::RocketQuat &m, const double *x, int B, bool warm)
    {
        return scpp_hip_scvx_setup(ctx, &m.p.abi, &opts, x, B, warm ? 1 : 0);
    }
    int scvxSetup(const scpp::models::Rocket2d &m, const double *x, int B, bool warm)
    {
        if (!m.p.constrain_initial_final)
            return SCPP_E_UNSUPPORTED; // model.info:55-56: "enable for SC and disable for MPC/LQR"
        const scpp_rocket2d_params a = m.abi();
        return scpp_hip_scvx_setup_rocket2d(ctx, &a, &opts, x, B, warm ? 1 : 0);
    }
    int batch_max, device, K_override;
    scpp_hip_ctx *ctx = nullptr;
    trajectory_data_t td;
    std::vector<trajectory_data_t> all_td;
    bool converged = false, recording = false;
    int iterations = 0, last_B = 0;
};

// commonFunctions.cpp:6-19
inline Model::input_vector_t interpolatedInput(const std::vector<Model::input_vector_t> &U, double t, double total_time,
                                               bool first_order_hold)
{
    const size_t K = U.size();
    const double time_step = total_time / double(K - 1);
    const size_t i = std::min(size_t(t / time_step), K - 2);
    const Model::input_vector_t u0 = U.at(i);
    const Model::input_vector_t u1 = first_order_hold ? U.at(i + 1) : u0;
    const double t_intermediate = std::fmod(t, time_step) / time_step;
    Model::input_vector_t u;
    for (size_t j = 0; j < size_t(Model::input_dim); j++)
        u[j] = u0[j] + (u1[j] - u0[j]) * t_intermediate;
    return u;
}

} // namespace scpp
