// mpc_sim: the reference's MPC_sim executable (scpp/src/MPC_sim.cpp:16-129) on the device engine.
//   default: ONE closed loop stepped from the host exactly like the reference's main loop (setInitialState, solve, simulate
//            under the previous input, u = U[0], stop when |x - x_final| < 0.02 or 15 s), every state recorded and the
//            reference's reduced output written to <out>/output/Rocket2D/MPC/<time>/0/{X,U,t}.txt;
//   --batch B: B closed loops with randomised start states, run entirely on the device (scpp_hip_mpc_sim), statistics only.
// The plant step is the constant 10 ms (the reference uses max(measured solve time, 10 ms)).
//   --batch B --seed S --steps n --config DIR --out DIR --device d --keep-constraint (do not clear constrain_initial_final)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mpc_algorithm.hpp"
#include "output.hpp"

namespace fs = std::filesystem;
using Model = scpp::models::Rocket2d;

// scpp/include/commonFunctions.hpp:19-32 (reduce_vector): `steps` samples at the integer stride size / steps -- a record
// shorter than `steps` therefore repeats its first entry, exactly as the reference's output does
template <class T>
static std::vector<T> thinOut(const std::vector<T> &record, size_t steps)
{
    const size_t stride = record.size() / steps;
    std::vector<T> kept;
    kept.reserve(steps);
    for (size_t pick = 0, at = 0; pick < steps; pick++, at += stride)
        kept.push_back(record.at(at));
    return kept;
}

int main(int argc, char **argv)
{
    std::string config = "../scpp_amd/config", out = "..";
    int batch = 0, device = 0, max_steps = 0;
    unsigned long long seed = 20260927ull;
    bool keep = false;
    for (int i = 1; i < argc; i++)
    {
        auto next = [&]() -> const char * {
            if (i + 1 >= argc)
            {
                std::fprintf(stderr, "missing value for %s\n", argv[i]);
                std::exit(2);
            }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--batch"))
            batch = std::atoi(next());
        else if (!std::strcmp(argv[i], "--seed"))
            seed = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--steps"))
            max_steps = std::atoi(next());
        else if (!std::strcmp(argv[i], "--config"))
            config = next();
        else if (!std::strcmp(argv[i], "--out"))
            out = next();
        else if (!std::strcmp(argv[i], "--device"))
            device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--keep-constraint"))
            keep = true;
        else
        {
            std::fprintf(stderr, "unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    try
    {
        Model::setParameterFolder(config);
        auto model = std::make_shared<Model>();
        model->loadParameters();
        if (!keep)
            model->p.constrain_initial_final = false; // model.info: "enable for SC and disable for MPC/LQR"
        const double sim_time = 15., min_timestep = 0.010;
        const size_t write_steps = 30;
        scpp::MPCAlgorithm solver(model, batch > 0 ? batch : 1, device);
        solver.initialize();
        solver.setFinalState(model->p.x_final);

        if (batch > 0)
        {
            const size_t B = size_t(batch);
            std::vector<double> xs(B * 6), xf(B * 6);
            for (size_t b = 0; b < B; b++)
            {
                const auto x = model->randomizedInitialState(seed, b);
                for (size_t i = 0; i < 6; i++)
                {
                    xs[b * 6 + i] = x[i];
                    xf[b * 6 + i] = model->p.x_final[i];
                }
            }
            int n_reached = 0;
            const auto t0 = std::chrono::steady_clock::now();
            if (scpp_hip_mpc_sim(solver.ctx, xs.data(), xf.data(), batch, min_timestep, sim_time, 0.02, max_steps, &n_reached) != SCPP_OK)
                throw std::runtime_error("scpp_hip_mpc_sim failed");
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::vector<int32_t> steps(B), failed(B);
            scpp_hip_mpc_sim_download(solver.ctx, nullptr, nullptr, nullptr, steps.data(), failed.data(), nullptr, nullptr);
            long total = 0, nf = 0;
            for (size_t b = 0; b < B; b++)
            {
                total += steps[b];
                nf += failed[b];
            }
            std::printf("Runtime: %.2fs\n", secs);
            std::printf("Average frequency: %.2fHz (%zu closed loops, %ld controller steps, %ld failed solves, %d reached the target)\n",
                        double(total) / secs, B, total, nf, n_reached);
            return 0;
        }

        Model::state_vector_t x = model->p.x_init;
        Model::input_vector_t u{0., 0.};
        scpp::MPCAlgorithm::state_vector_v_t X, X_sim;
        scpp::MPCAlgorithm::input_vector_v_t U, U_sim;
        std::vector<std::array<double, 1>> t_sim;
        double t = 0.;
        size_t sim_step = 0, failed = 0;
        const auto t_run = std::chrono::steady_clock::now();
        while (t < sim_time && (max_steps <= 0 || sim_step < size_t(max_steps)))
        {
            solver.setInitialState(x);
            const int ok = solver.solve();
            scpp::MPCAlgorithm::state_vector_v_t xv{x};
            solver.simulateBatch(min_timestep, {u}, {u}, xv);
            x = xv[0];
            t += min_timestep;
            if (ok)
            {
                solver.getSolution(X, U);
                u = U.at(0);
            }
            else
                failed++; // the previous input is held
            X_sim.push_back(x);
            U_sim.push_back(u);
            t_sim.push_back({t});
            sim_step++;
            double d2 = 0.;
            for (size_t i = 0; i < 6; i++)
                d2 += (x[i] - model->p.x_final[i]) * (x[i] - model->p.x_final[i]);
            if (std::sqrt(d2) < 0.02)
                break;
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run).count();
        std::printf("Runtime: %.2fs\nSimulated time: %.2fs\nAverage frequency: %.2fHz\nFailed solves: %zu\n", secs, t, double(sim_step) / t, failed);
        std::printf("Final state: %.6f %.6f %.6f %.6f %.6f %.6f\n", x[0], x[1], x[2], x[3], x[4], x[5]);

        const fs::path outputPath = fs::path(out) / "output" / Model::getModelName() / "MPC" / scpp::getTimeString() / "0";
        scpp::makeDir(outputPath);
        scpp::writeRows(outputPath / "X.txt", thinOut(X_sim, write_steps));
        scpp::writeRows(outputPath / "U.txt", thinOut(U_sim, write_steps));
        scpp::writeRows(outputPath / "t.txt", thinOut(t_sim, write_steps));
        std::printf("output: %s\n", outputPath.string().c_str());
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "mpc_sim: %s\n", e.what());
        return 1;
    }
    return 0;
}
