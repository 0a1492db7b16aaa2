// Microbenchmark: the memory access pattern of the matrix-pipe step kernels without the arithmetic.
// A wave streams down a column strip of a (n0 x n1) fp64 array in tiles of 16 rows; one load instruction covers 4 rows x
// 16 lanes x W bytes (W = 8: 128-byte row pieces, the B-operand layout of v_mfma_f64_16x16x4; W = 16: two interleaved
// strips, 256-byte row pieces).  PF tiles are in flight per wave.  Prints GB/s (read + write) for 64 arrays of 512 x 512.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>

template <int W, int PF, bool STORE>
__global__ __launch_bounds__(256) void strip_kernel(const double *src, double *dst, int n0, int n1, int S, int nseg, long long cstride) {
    using V = std::conditional_t<W == 16, double2, double>;
    constexpr int CW = W / 8 * 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int tj = blockIdx.x / nseg, seg = blockIdx.x - tj * nseg;
    const int col = (tj * 4 + wv) * CW + c * (W / 8);
    const int i_lo = seg * S, i_hi = min(n0, i_lo + S);
    const char *s = (const char *)(src + blockIdx.y * cstride + col);
    char *d = (char *)(dst + blockIdx.y * cstride + col);
    const unsigned pitch = n1 * 8;
    V buf[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) buf[u][q] = *(const V *)(s + (unsigned)min(i_lo + u * 16 + g + 4 * q, n0 - 1) * pitch);
    double acc = 0.0;
    for (int i0 = i_lo; i0 < i_hi; i0 += PF * 16) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = i0 + u * 16;
            V x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = buf[u][q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (W == 16) { x[q].x *= 1.0000001; x[q].y *= 1.0000001; } else x[q] *= 1.0000001;
                if (STORE) *(V *)(d + (unsigned)(i + g + 4 * q) * pitch) = x[q];
                else { if constexpr (W == 16) acc += x[q].x + x[q].y; else acc += x[q]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) buf[u][q] = *(const V *)(s + (unsigned)min(i + PF * 16 + g + 4 * q, n0 - 1) * pitch);
        }
    }
    if (!STORE && acc == 1.2345) dst[0] = acc;
}

template <int W, int PF, bool STORE>
static void run(double *buf, int nbuf, int chains, int n0, int n1, int S) {
    constexpr int CW = W / 8 * 16;
    const int nseg = (n0 + S - 1) / S, ncb = n1 / (4 * CW);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(ncb * nseg, chains);
    const size_t bs = (size_t)chains * n0 * n1;
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((strip_kernel<W, PF, STORE>), grid, dim3(256), 0, 0, buf + (k % nbuf) * bs, buf + ((k + 1) % nbuf) * bs, n0, n1, S, nseg, (long long)n0 * n1);
    hipDeviceSynchronize();
    const int reps = 32;
    hipEventRecord(e0);
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((strip_kernel<W, PF, STORE>), grid, dim3(256), 0, 0, buf + (k % nbuf) * bs, buf + ((k + 1) % nbuf) * bs, n0, n1, S, nseg, (long long)n0 * n1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)chains * n0 * n1 * 8.0 * (STORE ? 2 : 1);
    printf("W=%2d PF=%d store=%d S=%4d blocks=%5d : %7.1f us  %6.0f GB/s\n", W, PF, (int)STORE, S, grid.x * grid.y, us, bytes / us * 1e-3);
}

int main() {
    const int chains = 64, n0 = 512, n1 = 512;
    // step k reads buffer k, writes buffer k + 1 (the forward pass: each step reads what the previous launch wrote)
    const int nbuf = 16; double *src;
    hipMalloc(&src, sizeof(double) * nbuf * chains * n0 * n1);
    hipMemset(src, 0, sizeof(double) * nbuf * chains * n0 * n1);
    for (int S : {128, 256, 512}) {
        run<8, 2, true>(src, nbuf, chains, n0, n1, S);  run<8, 4, true>(src, nbuf, chains, n0, n1, S);
        run<16, 2, true>(src, nbuf, chains, n0, n1, S); run<16, 4, true>(src, nbuf, chains, n0, n1, S);
        run<8, 2, false>(src, nbuf, chains, n0, n1, S); run<8, 4, false>(src, nbuf, chains, n0, n1, S);
        run<16, 2, false>(src, nbuf, chains, n0, n1, S); run<16, 4, false>(src, nbuf, chains, n0, n1, S);
    }
    return 0;
}
