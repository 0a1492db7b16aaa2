// Microbenchmark: fp64 FMA issue rate / dependent latency, LDS read latency and shader clock on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CHAINS>
__global__ void fma_kernel(double *out, long long *cyc, long long *wall, int iters) {
    double a[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) a[k] = threadIdx.x * 1e-9 + k;
    const double m = 1.0000001, c = 1e-12;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) a[k] = fma(a[k], m, c);
    }
    long long t1 = clock64(), w1 = wall_clock64();
    double s = 0; 
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; wall[0] = w1 - w0; }
}

__global__ void lds_kernel(double *out, long long *cyc, int iters) {
    __shared__ double buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = i;
    __syncthreads();
    int idx = threadIdx.x;
    double s = 0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) { double v = buf[idx & 4095]; idx = (int)v + 1; s += v; }   // dependent reads
    long long t1 = clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; long long *cyc, *wall;
    CHECK(hipMalloc(&out, 1 << 24)); CHECK(hipMalloc(&cyc, 64)); CHECK(hipMalloc(&wall, 64));
    long long hc, hw;
    const int iters = 20000;
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, max shader clock %d kHz\n", wall_khz, clk_khz);
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 256 * waves_per_simd;   // one block per CU
        for (int chains : {1, 2, 4, 8}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int blocks) {
                if (chains == 1) hipLaunchKernelGGL(fma_kernel<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, wall, iters);
                if (chains == 2) hipLaunchKernelGGL(fma_kernel<2>, dim3(blocks), dim3(threads), 0, 0, out, cyc, wall, iters);
                if (chains == 4) hipLaunchKernelGGL(fma_kernel<4>, dim3(blocks), dim3(threads), 0, 0, out, cyc, wall, iters);
                if (chains == 8) hipLaunchKernelGGL(fma_kernel<8>, dim3(blocks), dim3(threads), 0, 0, out, cyc, wall, iters);
            };
            launch(256); hipDeviceSynchronize();
            hipEventRecord(e0); launch(256); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost);
            const double fmas = (double)iters * chains;
            printf("waves/SIMD %d chains %d: %.2f clk64-cycles per FMA per wave, shader MHz (clk64/wall) %.0f, kernel %.3f ms, chip TFLOPs %.1f\n",
                   waves_per_simd, chains, hc / fmas, (double)hc / ((double)hw / (wall_khz * 1e3)) / 1e6, ms,
                   2.0 * fmas * 256.0 * threads / (ms * 1e-3) / 1e12);
        }
    }
    hipLaunchKernelGGL(lds_kernel, dim3(1), dim3(64), 0, 0, out, cyc, 2000); hipDeviceSynchronize();
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent ds_read_b64 + cvt + add: %.1f cycles per iteration\n", hc / 2000.0);
    return 0;
}
