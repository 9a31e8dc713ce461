// CPU exercise of gather_layout.hpp with the semantics of ncclAllGather inside one ncclGroup per chunk (scvx_multi_gpu.cpp): every
// "device" stages its shard (zero-padded to the largest shard), the collective of chunk c copies rank g's rows of the chunk to
// gathered[chunk_off[c] + (g * count + r) * rowd] on every device, and reindexGathered must hand back every instance's row.
//   usage: gather_layout_test <batch> <gpus> <rowd> <chunk_mb>     exit 0 = every row of every device's copy is in place
#include <cstdio>
#include <cstdlib>

#include "gather_layout.hpp"

int main(int argc, char **argv)
{
    if (argc < 5)
        return 2;
    const int batch = std::atoi(argv[1]), gpus = std::atoi(argv[2]), rowd = std::atoi(argv[3]);
    const double chunk_mb = std::atof(argv[4]);
    const scpp::GatherLayout L = scpp::makeGatherLayout(batch, gpus, rowd, chunk_mb);
    auto value = [&](int inst, int e) { return double(inst) * 1000. + double(e) + 0.5; };
    std::vector<std::vector<double>> stage(size_t(gpus), std::vector<double>(size_t(L.nmax) * rowd, 0.));
    for (int g = 0; g < gpus; g++)
        for (int r = 0; r < L.shardSize(g); r++)
            for (int e = 0; e < rowd; e++)
                stage[size_t(g)][size_t(r) * rowd + e] = value(L.lo[size_t(g)] + r, e);
    int bad = 0;
    for (int dev = 0; dev < gpus; dev++) // every device receives the same buffer; check each one's copy
    {
        std::vector<double> gathered(L.gathered_doubles, -1.);
        for (size_t ci = 0; ci < L.chunks.size(); ci++)
            for (int g = 0; g < gpus; g++) // ncclAllGather(sendbuff = stage_g + first * rowd, count = count * rowd)
                std::memcpy(&gathered[L.chunk_off[ci] + size_t(g) * L.chunks[ci].second * rowd], &stage[size_t(g)][L.chunks[ci].first * rowd],
                            L.chunks[ci].second * rowd * sizeof(double));
        std::vector<double> all(size_t(batch) * rowd, -2.);
        scpp::reindexGathered(L, gathered.data(), all.data());
        for (int b = 0; b < batch; b++)
            for (int e = 0; e < rowd; e++)
                bad += all[size_t(b) * rowd + e] != value(b, e);
    }
    int covered = 0;
    for (int g = 0; g < gpus; g++)
        covered += L.shardSize(g);
    std::printf("batch %d gpus %d rowd %d: shards", batch, gpus, rowd);
    for (int g = 0; g < gpus; g++)
        std::printf(" %d", L.shardSize(g));
    std::printf(", nmax %d, %zu chunk(s) of <= %zu rows, receive buffer %zu doubles, wrong entries %d\n", L.nmax, L.chunks.size(), L.rows_per,
                L.gathered_doubles, bad);
    return (bad == 0 && covered == batch && L.lo[size_t(gpus)] == batch) ? 0 : 1;
}
