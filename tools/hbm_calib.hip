// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access widths ipm_kernel uses
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide reads by 2x, other widths are uncalibrated).
// Each kernel streams a buffer far larger than the 256 MB Infinity Cache exactly once with a known byte count;
// tools/pmc_hbm.sh divides the counter by that count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x2 __attribute__((vector_size(8)));
typedef unsigned int u32x4 __attribute__((vector_size(16)));

// 8 B per lane, one 512-byte row per wave-instruction (the field-major record rows of ipm_kernel)
__global__ void calib_read8(const double *p, size_t n, double *out)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    double acc = 0.;
    for (; i < n; i += stride)
        acc += p[i];
    if (acc == 1.2345e300)
        out[0] = acc;
}
// 16 B per lane
__global__ void calib_read16(const double2 *p, size_t n, double *out)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    double acc = 0.;
    for (; i < n; i += stride)
    {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300)
        out[0] = acc;
}
__global__ void calib_write8(double *p, size_t n)
{
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride)
        p[i] = double(i);
}
// 400-byte rows at a 400-byte pitch (K = 50 lanes of 64 active, rows not line-aligned): the record rows as they really are
__global__ void calib_read8_rows50(const double *p, size_t rows, double *out)
{
    const int lane = threadIdx.x & 63;
    size_t r = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const size_t stride = (size_t(gridDim.x) * blockDim.x) >> 6;
    double acc = 0.;
    for (; r < rows; r += stride)
        if (lane < 50)
            acc += p[r * 50 + lane];
    if (acc == 1.2345e300)
        out[0] = acc;
}
__global__ void calib_write8_rows50(double *p, size_t rows)
{
    const int lane = threadIdx.x & 63;
    size_t r = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const size_t stride = (size_t(gridDim.x) * blockDim.x) >> 6;
    for (; r < rows; r += stride)
        if (lane < 50)
            p[r * 50 + lane] = double(r);
}

int main()
{
    const size_t bytes = size_t(4) << 30; // 4 GiB >> 256 MB L3
    double *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess)
        return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const size_t n = bytes / 8;
    const dim3 grid(256 * 32), block(256);
    hipLaunchKernelGGL(calib_read8, grid, block, 0, 0, buf, n, out);
    hipLaunchKernelGGL(calib_read16, grid, block, 0, 0, (const double2 *)buf, n / 2, out);
    hipLaunchKernelGGL(calib_write8, grid, block, 0, 0, buf, n);
    const size_t rows = n / 50;
    hipLaunchKernelGGL(calib_read8_rows50, grid, block, 0, 0, buf, rows, out);
    hipLaunchKernelGGL(calib_write8_rows50, grid, block, 0, 0, buf, rows);
    hipDeviceSynchronize();
    printf("calib bytes %zu rows50_bytes %zu\n", bytes, rows * 400);
    return 0;
}
