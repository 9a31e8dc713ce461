// Register layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, determined by experiment: for every (lane of A, lane of B) pair with a unit entry
// the lanes of D that receive the product are printed -> the index maps A: lane -> (block, i, k), B: lane -> (block, k, j), D: lane -> (block, i, j).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main()
{
    double *a, *b, *d;
    hipMalloc(&a, 64 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&d, 64 * 8);
    double ha[64], hb[64], hd[64];
    // A = lane index + 1 (distinct primes would be nicer; use powers to decode): a[l] = 1 for one lane at a time, b = all ones
    for (int la = 0; la < 64; la++)
    {
        for (int i = 0; i < 64; i++) { ha[i] = (i == la) ? 1. : 0.; hb[i] = 1.; }
        hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
        k<<<1, 64>>>(a, b, d);
        hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
        printf("A lane %2d -> D lanes:", la);
        for (int i = 0; i < 64; i++) if (hd[i] != 0.) printf(" %d", i);
        printf("\n");
    }
    for (int lb = 0; lb < 64; lb++)
    {
        for (int i = 0; i < 64; i++) { hb[i] = (i == lb) ? 1. : 0.; ha[i] = 1.; }
        hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
        k<<<1, 64>>>(a, b, d);
        hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
        printf("B lane %2d -> D lanes:", lb);
        for (int i = 0; i < 64; i++) if (hd[i] != 0.) printf(" %d", i);
        printf("\n");
    }
    // which (A lane, B lane) pairs meet: A lane 0 with each B lane
    for (int la = 0; la < 64; la += 21)
    {
        printf("A lane %d meets B lanes:", la);
        for (int lb = 0; lb < 64; lb++)
        {
            for (int i = 0; i < 64; i++) { ha[i] = (i == la) ? 1. : 0.; hb[i] = (i == lb) ? 1. : 0.; }
            hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
            k<<<1, 64>>>(a, b, d);
            hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
            for (int i = 0; i < 64; i++) if (hd[i] != 0.) printf(" %d(D%d)", lb, i);
        }
        printf("\n");
    }
    return 0;
}
