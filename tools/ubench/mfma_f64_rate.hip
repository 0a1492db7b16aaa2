// Microbenchmark: v_mfma_f64_16x16x4_f64 rate on gfx950, alone and interleaved with fp64 VALU FMAs; plus a layout check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU>
__global__ void mfma_kernel(double *out, int iters) {
    d4 acc[NACC > 0 ? NACC : 1];
    for (int k = 0; k < NACC; ++k) acc[k] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double v[NVALU > 0 ? NVALU : 1];
    for (int k = 0; k < NVALU; ++k) v[k] = threadIdx.x * 1e-9 + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NVALU; ++k) v[k] = fma(v[k], 1.0000001, 1e-12);
    }
    double s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    for (int k = 0; k < NVALU; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void layout_kernel(const double *A, const double *B, double *D) {   // A 16x4 row-major, B 4x16 row-major, D 16x16
    const int l = threadIdx.x, c = l & 15, g = l >> 4;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c * 4 + g], B[g * 16 + c], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(g + 4 * r) * 16 + c] = acc[r];
}

// MFMA f64 next to 32-bit integer VALU work (does the f64 matrix op leave issue slots to the integer ALU?)
template <int NACC, int NINT>
__global__ void mfma_int_kernel(double *out, int iters) {
    d4 acc[NACC > 0 ? NACC : 1];
    for (int k = 0; k < NACC; ++k) acc[k] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    unsigned v[NINT > 0 ? NINT : 1];
    for (int k = 0; k < NINT; ++k) v[k] = threadIdx.x + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NINT; ++k) v[k] = (v[k] ^ (unsigned)i) + 0x9e3779b9u;
    }
    double s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    unsigned t = 0;
    for (int k = 0; k < NINT; ++k) t += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + t;
}

template <int NACC, int NINT>
static void run_int(double *out, int waves, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves;
    hipLaunchKernelGGL((mfma_int_kernel<NACC, NINT>), dim3(256), dim3(threads), 0, 0, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_int_kernel<NACC, NINT>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD %d  MFMA %d + 2x%d int VALU ops per iter: %.1f ns per iteration per wave-slot (MFMA alone would be %.1f ns at 77 TFLOP/s)\n",
           waves, NACC, NINT, ms * 1e6 / iters / waves, NACC * 2048.0 * 256 * 4 / 77e12 * 1e9);
}

template <int NACC, int NVALU>
static void run(double *out, int waves, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves;
    hipLaunchKernelGGL((mfma_kernel<NACC, NVALU>), dim3(256), dim3(threads), 0, 0, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_kernel<NACC, NVALU>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * NACC * 256.0 * 4 * waves, nv = (double)iters * NVALU * 256.0 * 4 * waves;
    printf("waves/SIMD %d  MFMA chains %d + VALU fma %d per iter: %.3f ms  MFMA %.1f TFLOP/s  VALU %.1f TFLOP/s  (%.1f ns per iteration)\n",
           waves, NACC, NVALU, ms, nm * 2048 / (ms * 1e-3) / 1e12, nv * 128 / (ms * 1e-3) / 1e12, ms * 1e6 / iters);
}

int main() {
    double *out; hipMalloc(&out, 1 << 24);
    // layout check: asymmetric A, B
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11 + (i % 7); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]) / fabs(R[i]));
    printf("layout check (A row=lane&15,k=lane>>4; B k=lane>>4,col=lane&15; D row=(lane>>4)+4r, col=lane&15): max rel err %.2e\n", err);
    const int it = 20000;
    for (int w : {1, 2, 4}) { run<1, 0>(out, w, it); run<2, 0>(out, w, it); run<4, 0>(out, w, it); }
    for (int w : {1, 2, 4}) { run<0, 8>(out, w, it); run<0, 16>(out, w, it); }
    for (int w : {1, 2, 4}) { run<2, 8>(out, w, it); run<2, 16>(out, w, it); run<2, 32>(out, w, it); run<4, 16>(out, w, it); }
    for (int w : {1, 2}) { run_int<0, 32>(out, w, it); run_int<2, 0>(out, w, it); run_int<2, 16>(out, w, it); run_int<2, 32>(out, w, it); run_int<2, 64>(out, w, it); }
    return 0;
}
