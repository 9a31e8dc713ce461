// Probe of the gfx950 row-swap instructions the elimination relies on (common.h: rowGroupBcast / rowGroupDiag).
// hipcc --offload-arch=gfx950 -O2 tests/tools/permlane_probe.hip -o tools/bin/permlane_probe ; prints PASS / FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../scpp_amd/csrc/common.h"
__global__ void probe(double *out)
{
    const int l = threadIdx.x;
    const double v = 1000. + l;
    out[l] = scpp::rowGroupBcast<0>(v);
    out[64 + l] = scpp::rowGroupBcast<1>(v);
    out[128 + l] = scpp::rowGroupBcast<2>(v);
    out[192 + l] = scpp::rowGroupBcast<3>(v);
    out[256 + l] = scpp::rowGroupDiag(v);
}
int main()
{
    double *d, h[320];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int gs = 0; gs < 4; gs++)
        for (int l = 0; l < 64; l++)
            bad += h[gs * 64 + l] != 1000. + gs * 16 + (l & 15);
    for (int l = 0; l < 64; l++)
        bad += h[256 + l] != 1000. + (l & 3) * 16 + (l & 15);
    std::printf("permlane probe: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    if (bad)
        for (int l = 0; l < 64; l++)
            std::printf("lane %2d: b0 %g b1 %g b2 %g b3 %g diag %g\n", l, h[l], h[64 + l], h[128 + l], h[192 + l], h[256 + l]);
    return bad ? 1 : 0;
}
