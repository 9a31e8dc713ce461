// sc_oneshot: the reference's SC_oneshot executable (scpp/src/SC_oneshot.cpp:15-64) on the device engine.
//   no arguments      : one trajectory from the shipped configuration, every iterate written to
//                       <out>/output/RocketQuat/SC/<time>/<iter>/{X,U,t}.txt   (what the reference does)
//   --batch B [--seed S]: B randomised initial states solved at once (the workload of BASELINE configs 1/2);
//                       instance 0 is written to the same tree under iteration index 0, a summary goes to stdout
//   --scvx            : run the SCvx variant (SCvxAlgorithm, SCvx.info) instead of SCAlgorithm; output under .../SCvx/
//   --gpus N          : with --batch, shard the instances over N devices (contiguous static shards, one host thread + one context
//                       per GPU, nothing shared while solving -- SURVEY 8(e)); results are concatenated in instance order
//   --config DIR --out DIR --K n --device d
// The active model is a compile definition like in the reference (CMakeLists.txt:33-55): `sc_oneshot` is built for RocketQuat,
// `sc_oneshot_rocket2d` (-DSCPP_ACTIVE_MODEL_ROCKET2D) for Rocket2d, the reference's default (activeModel.hpp:10).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "output.hpp"
#include "sc_algorithm.hpp"

namespace fs = std::filesystem;

int main(int argc, char **argv)
{
    std::string config = "../scpp_amd/config", out = "..";
    int batch = 0, K = 0, device = 0, gpus = 1;
    unsigned long long seed = 20260927ull;
    bool scvx = false;
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
        if (!std::strcmp(argv[i], "--scvx"))
            scvx = true;
        else if (!std::strcmp(argv[i], "--batch"))
            batch = std::atoi(next());
        else if (!std::strcmp(argv[i], "--seed"))
            seed = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--config"))
            config = next();
        else if (!std::strcmp(argv[i], "--out"))
            out = next();
        else if (!std::strcmp(argv[i], "--K"))
            K = std::atoi(next());
        else if (!std::strcmp(argv[i], "--device"))
            device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--gpus"))
            gpus = std::atoi(next());
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

        std::vector<trajectory_data_t> all_td;
        if (scvx)
        {
            scpp::SCvxAlgorithm vsolver(model, batch > 0 ? batch : 1, device, K);
            vsolver.initialize();
            std::vector<Model::state_vector_t> x_inits;
            if (batch <= 0)
                x_inits.push_back(model->p.x_init);
            for (int b = 0; b < batch; b++)
            {
                Model inst = *model;
                inst.p.randomizeInitialState(seed, uint64_t(b));
                x_inits.push_back(inst.p.x_init);
            }
            scpp::scvx_result_t r;
            vsolver.recordIterates(true); // getAllSolutions: every iterate of instance 0 is written below, like SC_oneshot.cpp:31-63 does for all_td
            vsolver.solveBatch(x_inits, r);
            long conv = 0, fails = 0, iters = 0, solves = 0;
            for (size_t b = 0; b < x_inits.size(); b++)
            {
                conv += r.converged[b];
                fails += r.status[b] != 0;
                iters += r.sc_iterations[b];
                solves += r.solves[b];
            }
            const double n = double(x_inits.size());
            std::printf("SCvx batch %zu: converged %ld, solver failures %ld, mean iterations %.2f, mean sub-problem solves %.2f\n",
                        x_inits.size(), conv, fails, double(iters) / n, double(solves) / n);
            // output/<Model>/SCvx/<time>/<iteration>/{X,U,t}.txt for every trajectory of all_td (SC_oneshot.cpp:31-63: iteration 0 is the initial
            // trajectory, iteration j the trajectory after the j-th iteration), instance 0 of the batch
            std::vector<std::vector<trajectory_data_t>> all;
            vsolver.getAllSolutionsBatch(all);
            all_td = all.at(0);
            const fs::path outputPath = fs::path(out) / "output" / Model::getModelName() / "SCvx" / scpp::getTimeString();
            for (size_t k = 0; k < all_td.size(); k++)
            {
                const fs::path iterationPath = outputPath / std::to_string(k);
                scpp::makeDir(iterationPath);
                scpp::writeRows(iterationPath / "X.txt", all_td[k].X);
                scpp::writeRows(iterationPath / "U.txt", all_td[k].U);
                std::ofstream f(iterationPath / "t.txt");
                f << all_td[k].t;
            }
            std::printf("output: %s (%zu iterates)\n", outputPath.string().c_str(), all_td.size());
            return 0;
        }
        if (batch > 0 && gpus > 1)
        {
            // one host thread + one device context per GPU; shard g owns instances [lo_g, hi_g)
            std::vector<Model::state_vector_t> x_inits;
            for (int b = 0; b < batch; b++)
            {
                Model inst = *model;
                inst.p.randomizeInitialState(seed, uint64_t(b));
                x_inits.push_back(inst.p.x_init);
            }
            std::vector<scpp::batch_result_t> part(static_cast<size_t>(gpus));
            std::vector<std::string> err(static_cast<size_t>(gpus));
            std::vector<std::thread> th;
            const int base = batch / gpus, rem = batch % gpus;
            for (int g = 0; g < gpus; g++)
            {
                const int lo = g * base + std::min(g, rem), hi = lo + base + (g < rem ? 1 : 0);
                th.emplace_back([&, g, lo, hi]() {
                    try
                    {
                        if (hi <= lo)
                            return;
                        scpp::SCAlgorithm shard(model, hi - lo, device + g, K);
                        shard.initialize();
                        const std::vector<Model::state_vector_t> xs(x_inits.begin() + lo, x_inits.begin() + hi);
                        shard.solveBatch(xs, part[size_t(g)]);
                    }
                    catch (const std::exception &e)
                    {
                        err[size_t(g)] = e.what();
                    }
                });
            }
            for (auto &t : th)
                t.join();
            long conv = 0, fails = 0, iters = 0, n = 0;
            for (int g = 0; g < gpus; g++)
            {
                if (!err[size_t(g)].empty())
                    throw std::runtime_error("GPU " + std::to_string(device + g) + ": " + err[size_t(g)]);
                const scpp::batch_result_t &r = part[size_t(g)];
                for (size_t b = 0; b < r.td.size(); b++, n++)
                {
                    conv += r.converged[b];
                    fails += r.status[b] != 0;
                    iters += r.sc_iterations[b];
                }
            }
            std::printf("batch %d on %d GPUs: converged %ld, solver failures %ld, mean SC iterations %.2f\n", batch, gpus, conv, fails,
                        double(iters) / double(n));
            // checksum of the concatenated result (tests compare it with the single-device run)
            double cs = 0.;
            for (int g = 0; g < gpus; g++)
                for (const auto &t : part[size_t(g)].td)
                    for (const auto &x : t.X)
                        for (double v : x)
                            cs += v;
            std::printf("checksum X %.12e\n", cs);
            all_td.push_back(part[0].td[0]);
            const fs::path outputPath = fs::path(out) / "output" / Model::getModelName() / "SC" / scpp::getTimeString() / "0";
            scpp::makeDir(outputPath);
            scpp::writeRows(outputPath / "X.txt", all_td[0].X);
            scpp::writeRows(outputPath / "U.txt", all_td[0].U);
            std::ofstream f(outputPath / "t.txt");
            f << all_td[0].t;
            std::printf("output: %s\n", outputPath.string().c_str());
            return 0;
        }
        scpp::SCAlgorithm solver(model, batch > 0 ? batch : 1, device, K);
        solver.initialize();

        if (batch <= 0)
        {
            solver.solve();
            solver.getAllSolutions(all_td);
            std::printf("%s after %d iterations.\n", solver.hasConverged() ? "Converged" : "No convergence", solver.getIterations());
        }
        else
        {
            std::vector<Model::state_vector_t> x_inits;
            for (int b = 0; b < batch; b++)
            {
                Model inst = *model;
                inst.p.randomizeInitialState(seed, uint64_t(b));
                x_inits.push_back(inst.p.x_init);
            }
            scpp::batch_result_t r;
            solver.solveBatch(x_inits, r);
            long conv = 0, fails = 0, iters = 0;
            for (int b = 0; b < batch; b++)
            {
                conv += r.converged[size_t(b)];
                fails += r.status[size_t(b)] != 0;
                iters += r.sc_iterations[size_t(b)];
            }
            std::printf("batch %d: converged %ld, solver failures %ld, mean SC iterations %.2f\n", batch, conv, fails, double(iters) / batch);
            double cs = 0.;
            for (const auto &t : r.td)
                for (const auto &x : t.X)
                    for (double v : x)
                        cs += v;
            std::printf("checksum X %.12e\n", cs);
            all_td.push_back(r.td[0]);
        }

        const fs::path outputPath = fs::path(out) / "output" / Model::getModelName() / "SC" / scpp::getTimeString();
        for (size_t k = 0; k < all_td.size(); k++)
        {
            const fs::path iterationPath = outputPath / std::to_string(k);
            scpp::makeDir(iterationPath);
            scpp::writeRows(iterationPath / "X.txt", all_td[k].X);
            scpp::writeRows(iterationPath / "U.txt", all_td[k].U);
            std::ofstream f(iterationPath / "t.txt");
            f << all_td[k].t;
        }
        std::printf("output: %s\n", outputPath.string().c_str());
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "sc_oneshot: %s\n", e.what());
        return 1;
    }
    return 0;
}
