// scvx_multi_gpu: the headline workload (converged SCvx trajectories, RocketQuat) driven from C++17 on N GPUs of one node, with the
// result rows all-gathered by RCCL over xGMI -- the host-side shape BASELINE.json's north_star names ("host C++17 calls HIP through a
// thin C-ABI shim; batches shard trivially across the GPUs of one node with an RCCL all-gather only to collect converged
// trajectories").  One PROCESS, one host thread + one scpp_hip context + one RCCL communicator per device (ncclCommInitAll):
//   shard g owns the contiguous instance ids [lo_g, hi_g) and runs scpp_hip_scvx_solve_stream on them (no collective in the loop);
//   the device-resident result rows (scpp_hip_stream_rows) are gathered with ncclAllGather in chunks of <= --chunk-mb per rank
//   (uneven shards are padded to the largest shard inside the chunk loop), after which every device holds every row;
//   device 0's copy is downloaded, checked against each shard's own rows (bitwise) and summarised.
// bench.py does the same through torch.distributed (one process per GPU); this is the torch-free form for a SCpp maintainer.
//   --batch B (instances) --gpus N --seed S --slots n (resident slots per GPU, 0 = shard size) --chunk-mb M --config DIR --K n
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "gather_layout.hpp"
#include "sc_algorithm.hpp"

#define CHECK(call)                                                                                      \
    do                                                                                                   \
    {                                                                                                    \
        const int rc_ = int(call);                                                                       \
        if (rc_ != 0)                                                                                    \
            throw std::runtime_error(std::string(#call) + " failed with code " + std::to_string(rc_));   \
    } while (0)

int main(int argc, char **argv)
{
    std::string config = "../scpp_amd/config";
    int batch = 64, gpus = 1, K = 0, slots = 0;
    double chunk_mb = 64.;
    unsigned long long seed = 20260927ull;
    for (int i = 1; i < argc; i++)
    {
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : (std::fprintf(stderr, "missing value\n"), std::exit(2), ""); };
        if (!std::strcmp(argv[i], "--batch"))
            batch = std::atoi(next());
        else if (!std::strcmp(argv[i], "--gpus"))
            gpus = std::atoi(next());
        else if (!std::strcmp(argv[i], "--seed"))
            seed = std::strtoull(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--slots"))
            slots = std::atoi(next());
        else if (!std::strcmp(argv[i], "--chunk-mb"))
            chunk_mb = std::atof(next());
        else if (!std::strcmp(argv[i], "--config"))
            config = next();
        else if (!std::strcmp(argv[i], "--K"))
            K = std::atoi(next());
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
        scpp::SCvxAlgorithm probe(model, 1, 0, K); // SCvx.info through the reference-shaped front end
        probe.loadParameters();
        const scpp_scvx_opts opts = probe.opts;
        const int Kn = opts.K, rowd = Kn * (int(Model::state_dim) + int(Model::input_dim)) + 10;

        std::vector<Model::state_vector_t> x_inits;
        for (int b = 0; b < batch; b++)
        {
            Model inst = *model;
            inst.p.randomizeInitialState(seed, uint64_t(b));
            x_inits.push_back(inst.p.x_init);
        }
        // shards, chunks and receive-buffer offsets of the one collective (gather_layout.hpp; CPU-tested with gpus > 1)
        const scpp::GatherLayout GL = scpp::makeGatherLayout(batch, gpus, rowd, chunk_mb);
        const int nmax = GL.nmax;
        const std::vector<int> &lo = GL.lo;

        std::vector<int> devs(static_cast<size_t>(gpus));
        for (int g = 0; g < gpus; g++)
            devs[size_t(g)] = g;
        std::vector<ncclComm_t> comms(static_cast<size_t>(gpus));
        CHECK(ncclCommInitAll(comms.data(), gpus, devs.data()));

        std::vector<scpp_hip_ctx *> ctx(size_t(gpus), nullptr);
        std::vector<double *> rows(size_t(gpus), nullptr), gathered(size_t(gpus), nullptr), stage(size_t(gpus), nullptr);
        std::vector<hipStream_t> streams(static_cast<size_t>(gpus));
        std::vector<int> nconv(size_t(gpus), 0);
        std::vector<std::string> err(static_cast<size_t>(gpus));
        const auto t0 = std::chrono::steady_clock::now();
        {
            std::vector<std::thread> th;
            for (int g = 0; g < gpus; g++)
                th.emplace_back([&, g]() {
                    try
                    {
                        const int n = lo[size_t(g) + 1] - lo[size_t(g)];
                        CHECK(hipSetDevice(g));
                        CHECK(hipStreamCreate(&streams[size_t(g)]));
                        // every device receives every row; the send side is staged so that all ranks contribute nmax rows
                        CHECK(hipMalloc(reinterpret_cast<void **>(&gathered[size_t(g)]), size_t(gpus) * size_t(nmax) * rowd * sizeof(double)));
                        CHECK(hipMalloc(reinterpret_cast<void **>(&stage[size_t(g)]), size_t(nmax) * rowd * sizeof(double)));
                        CHECK(hipMemset(stage[size_t(g)], 0, size_t(nmax) * rowd * sizeof(double)));
                        if (n <= 0)
                            return;
                        const int S = slots > 0 ? std::min(slots, n) : n;
                        CHECK(scpp_hip_create(&ctx[size_t(g)], g, Model::model_id, Kn, S, 0));
                        CHECK(scpp_hip_scvx_solve_stream(ctx[size_t(g)], &model->p.abi, &opts, &x_inits[size_t(lo[size_t(g)])][0], n, S, 0, &nconv[size_t(g)]));
                        void *r = nullptr;
                        int rd = 0, nr = 0;
                        CHECK(scpp_hip_stream_rows(ctx[size_t(g)], &r, &rd, &nr));
                        if (rd != rowd || nr != n)
                            throw std::runtime_error("unexpected row layout");
                        rows[size_t(g)] = static_cast<double *>(r);
                        CHECK(hipMemcpyAsync(stage[size_t(g)], r, size_t(n) * rowd * sizeof(double), hipMemcpyDeviceToDevice, streams[size_t(g)]));
                        CHECK(hipStreamSynchronize(streams[size_t(g)]));
                    }
                    catch (const std::exception &e)
                    {
                        err[size_t(g)] = e.what();
                    }
                });
            for (auto &t : th)
                t.join();
        }
        for (int g = 0; g < gpus; g++)
            if (!err[size_t(g)].empty())
                throw std::runtime_error("GPU " + std::to_string(g) + ": " + err[size_t(g)]);
        // ---- the one collective: all-gather of the result rows, <= chunk_mb per rank and call (xGMI is point-to-point: the
        //      receive buffer of a call is gpus x chunk).  Layout of `gathered` on every device: [chunk][rank][rows of the chunk]
        //      -> re-indexed on the host; each chunk is one ncclGroup over the local communicators.
        const size_t rows_per = GL.rows_per;
        int n_coll = 0;
        const std::vector<std::pair<size_t, size_t>> &chunks = GL.chunks; // (first row, count) within a shard
        for (size_t ci = 0; ci < chunks.size(); ci++)
        {
            const auto &c = chunks[ci];
            const size_t off = GL.chunk_off[ci];
            CHECK(ncclGroupStart());
            for (int g = 0; g < gpus; g++)
                CHECK(ncclAllGather(stage[size_t(g)] + c.first * rowd, gathered[size_t(g)] + off, c.second * rowd, ncclDouble, comms[size_t(g)],
                                    streams[size_t(g)]));
            CHECK(ncclGroupEnd());
            n_coll++;
        }
        for (int g = 0; g < gpus; g++)
        {
            CHECK(hipSetDevice(g));
            CHECK(hipStreamSynchronize(streams[size_t(g)]));
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // ---- device 0's gathered copy -> host, back into instance order; every shard's own rows must be in it bitwise ----
        std::vector<double> hg(size_t(gpus) * size_t(nmax) * rowd), all(size_t(batch) * rowd), own;
        CHECK(hipSetDevice(0));
        CHECK(hipMemcpy(hg.data(), gathered[0], hg.size() * sizeof(double), hipMemcpyDeviceToHost));
        scpp::reindexGathered(GL, hg.data(), all.data());
        bool same = true;
        for (int g = 0; g < gpus; g++)
        {
            const int n = lo[size_t(g) + 1] - lo[size_t(g)];
            if (n <= 0)
                continue;
            own.resize(size_t(n) * rowd);
            CHECK(scpp_hip_stream_download(ctx[size_t(g)], own.data(), 0, n));
            same = same && !std::memcmp(own.data(), &all[size_t(lo[size_t(g)]) * rowd], own.size() * sizeof(double));
        }
        long conv = 0, fails = 0;
        double cs = 0.;
        const int sc0 = Kn * (int(Model::state_dim) + int(Model::input_dim));
        for (int b = 0; b < batch; b++)
        {
            const double *row = &all[size_t(b) * rowd];
            conv += long(row[sc0 + 6]);
            fails += row[sc0 + 7] != 0.;
            if (long(row[sc0 + 9]) != long(b - lo[0]) && gpus == 1)
                same = false;
            for (int e = 0; e < Kn * int(Model::state_dim); e++)
                cs += row[e];
        }
        long cs_conv = 0;
        for (int g = 0; g < gpus; g++)
            cs_conv += nconv[size_t(g)];
        std::printf("SCvx %d instances on %d GPU(s): converged %ld (engine counters %ld), solver failures %ld, %.1f converged/s incl. set-up\n", batch,
                    gpus, conv, cs_conv, fails, double(conv) / secs);
        std::printf("RCCL all-gather: %d collective(s) of <= %zu rows (%.1f MB) per rank; gathered rows == shard rows bitwise: %s\n", n_coll, rows_per,
                    double(rows_per) * rowd * 8e-6, same ? "yes" : "NO");
        std::printf("checksum X %.12e\n", cs);
        for (int g = 0; g < gpus; g++)
        {
            if (ctx[size_t(g)])
                scpp_hip_destroy(ctx[size_t(g)]);
            ncclCommDestroy(comms[size_t(g)]);
        }
        return same && conv == cs_conv ? 0 : 1;
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "scvx_multi_gpu: %s\n", e.what());
        return 1;
    }
}
