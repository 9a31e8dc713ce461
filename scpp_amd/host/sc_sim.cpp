// sc_sim: the reference's SC_sim executable (scpp/src/SC_sim.cpp:19-104) on the device engine, for B independent
// closed loops (B = 1 without arguments: exactly the reference's run).  Per step: solve(warm_start = step > 0),
// u0 = U[0], u1 = interpolatedInput(U, dt, t), x <- simulate(dt, u0, u1, x) with x aliasing the next solve's x_init,
// stop per loop when ||x - x_final|| < 0.02 or t < 0.25.  Loop 0 is written to <out>/output/RocketQuat/SC_sim/<time>/0/.
//   --batch B --seed S --steps n --config DIR --out DIR --K n --device d
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "output.hpp"
#include "sc_algorithm.hpp"

namespace fs = std::filesystem;

int main(int argc, char **argv)
{
    std::string config = "../scpp_amd/config", out = "..";
    int batch = 1, K = 0, device = 0;
    size_t max_steps = 100;
    unsigned long long seed = 20260927ull;
    bool randomize = false;
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
        {
            batch = std::atoi(next());
            randomize = true;
        }
        else if (!std::strcmp(argv[i], "--seed"))
            seed = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--steps"))
            max_steps = size_t(std::atoi(next()));
        else if (!std::strcmp(argv[i], "--config"))
            config = next();
        else if (!std::strcmp(argv[i], "--out"))
            out = next();
        else if (!std::strcmp(argv[i], "--K"))
            K = std::atoi(next());
        else if (!std::strcmp(argv[i], "--device"))
            device = std::atoi(next());
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
        scpp::SCAlgorithm solver(model, batch, device, K);
        solver.initialize();

        const double time_step = 0.05;
        const size_t B = size_t(batch);
        std::vector<Model::state_vector_t> x(B, model->p.x_init);
        if (randomize)
            for (size_t b = 0; b < B; b++)
            {
                Model inst = *model;
                inst.p.randomizeInitialState(seed, uint64_t(b));
                x[b] = inst.p.x_init;
            }
        std::vector<int32_t> active(B, 1);
        std::vector<std::vector<Model::state_vector_t>> X_sim(B);
        std::vector<std::vector<Model::input_vector_t>> U_sim(B);
        std::vector<size_t> steps(B, 0);

        const auto t_run = std::chrono::steady_clock::now();
        size_t sim_step = 0, total_solves = 0;
        while (sim_step < max_steps)
        {
            size_t n_active = 0;
            for (auto a : active)
                n_active += size_t(a != 0);
            if (!n_active)
                break;
            const bool warm_start = sim_step > 0;
            scpp::batch_result_t r;
            solver.solveBatch(x, r, warm_start, &active);
            total_solves += n_active;
            std::vector<Model::input_vector_t> u0(B), u1(B);
            for (size_t b = 0; b < B; b++)
            {
                u0[b] = r.td[b].U.at(0);
                u1[b] = scpp::interpolatedInput(r.td[b].U, time_step, r.td[b].t, r.td[b].interpolatedInput());
            }
            std::vector<Model::state_vector_t> x_new = x;
            solver.simulateBatch(time_step, u0, u1, x_new);
            for (size_t b = 0; b < B; b++)
            {
                if (!active[b])
                    continue;
                if (r.status[b] != 0)
                {
                    active[b] = 0; // the reference would terminate (SCAlgorithm.cpp:94-98)
                    continue;
                }
                x[b] = x_new[b];
                X_sim[b].push_back(x[b]);
                U_sim[b].push_back(u0[b]);
                steps[b]++;
                double d2 = 0.;
                for (size_t j = 0; j < 14; j++)
                    d2 += (x[b][j] - model->p.x_final[j]) * (x[b][j] - model->p.x_final[j]);
                const bool reached_end = std::sqrt(d2) < 0.02 || r.td[b].t < 0.25;
                if (reached_end)
                    active[b] = 0;
            }
            sim_step++;
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run).count();
        std::printf("Time, %zu steps: %.2fms\n", sim_step, 1e3 * secs);
        std::printf("Average frequency: %.2fHz (%zu closed loops, %zu solves)\n", double(total_solves) / secs, B, total_solves);

        const fs::path outputPath = fs::path(out) / "output" / Model::getModelName() / "SC_sim" / scpp::getTimeString() / std::to_string(0);
        scpp::makeDir(outputPath);
        scpp::writeRows(outputPath / "X.txt", X_sim[0]);
        scpp::writeRows(outputPath / "U.txt", U_sim[0]);
        {
            std::ofstream f(outputPath / "t.txt");
            f << double(steps[0]) * time_step;
        }
        std::printf("output: %s\n", outputPath.string().c_str());
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "sc_sim: %s\n", e.what());
        return 1;
    }
    return 0;
}
