// Layout of the chunked RCCL all-gather of the result rows (scvx_multi_gpu.cpp) and the host-side re-indexing of the gathered
// buffer back into instance order -- factored out so that the arithmetic can be exercised without a GPU (gather_layout_test.cpp:
// gpus > 1, uneven shards, several chunks).  SURVEY 8(e): static contiguous shards, ONE collective kind.
//
//   shard g owns the instance ids [lo[g], lo[g+1]) (the first `batch % gpus` shards hold one instance more);
//   every rank contributes nmax = max shard size rows (short shards are zero-padded in their staging buffer);
//   the rows travel in chunks of <= rows_per rows per rank: chunk c = shard-local rows [first_c, first_c + count_c);
//   ncclAllGather of chunk c writes, on every device, rank g's count_c rows at
//       gathered[chunk_off[c] + (g * count_c + r) * rowd],   chunk_off[c] = sum_{c' < c} gpus * count_c' * rowd.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <utility>
#include <vector>

namespace scpp
{
struct GatherLayout
{
    int batch = 0, gpus = 1, rowd = 0, nmax = 0;
    size_t rows_per = 1;
    std::vector<int> lo;                            // [gpus + 1] first instance id of every shard
    std::vector<std::pair<size_t, size_t>> chunks;  // (first shard-local row, count)
    std::vector<size_t> chunk_off;                  // offset (doubles) of a chunk's receive block in `gathered`
    size_t gathered_doubles = 0;                    // size of the receive buffer of one device
    int shardSize(int g) const { return lo[size_t(g) + 1] - lo[size_t(g)]; }
};

inline GatherLayout makeGatherLayout(int batch, int gpus, int rowd, double chunk_mb)
{
    GatherLayout L;
    L.batch = batch;
    L.gpus = gpus;
    L.rowd = rowd;
    const int base = batch / gpus, rem = batch % gpus;
    L.nmax = base + (rem ? 1 : 0);
    L.lo.assign(size_t(gpus) + 1, 0);
    for (int g = 0; g < gpus; g++)
        L.lo[size_t(g) + 1] = L.lo[size_t(g)] + base + (g < rem ? 1 : 0);
    L.rows_per = std::max<size_t>(1, size_t(chunk_mb * 1e6) / (size_t(rowd) * sizeof(double)));
    size_t off = 0;
    for (size_t first = 0; first < size_t(L.nmax); first += L.rows_per)
    {
        const size_t count = std::min(L.rows_per, size_t(L.nmax) - first);
        L.chunks.emplace_back(first, count);
        L.chunk_off.push_back(off);
        off += size_t(gpus) * count * size_t(rowd);
    }
    L.gathered_doubles = off;
    return L;
}

// one device's gathered buffer -> rows in instance order (all[b * rowd ..]); padding rows of short shards are skipped
inline void reindexGathered(const GatherLayout &L, const double *gathered, double *all)
{
    for (size_t ci = 0; ci < L.chunks.size(); ci++)
        for (int g = 0; g < L.gpus; g++)
        {
            const int n = L.shardSize(g);
            for (size_t r = 0; r < L.chunks[ci].second; r++)
            {
                const size_t row = L.chunks[ci].first + r;
                if (row < size_t(n))
                    std::memcpy(&all[(size_t(L.lo[size_t(g)]) + row) * size_t(L.rowd)],
                                &gathered[L.chunk_off[ci] + (size_t(g) * L.chunks[ci].second + r) * size_t(L.rowd)], size_t(L.rowd) * sizeof(double));
            }
        }
}
} // namespace scpp
