// Microbenchmark (gfx950): v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 products per instruction, 256 FMAs) against
// v_mfma_f64_16x16x4_f64 (1024 FMAs): time per instruction with 1 / 2 waves per SIMD and 1 / 4 independent accumulator chains,
// plus the operand layout of the 4-block form.   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_4x4.hip -o /tmp/m44 && /tmp/m44
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SMALL>
__global__ void k(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double s = 0;
    if (SMALL) {
        double acc[NACC];
        for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < NACC; ++q) s += acc[q];
    } else {
        d4 acc[NACC];
        for (int q = 0; q < NACC; ++q) acc[q] = d4{0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < NACC; ++q) s += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// layout probe: A[l], B[l] one double per lane; D one double per lane.  With A = one-hot at lane la and B = one-hot at lane lb,
// which D lanes become non-zero?  -> (block, i, k) of A-lane, (block, k, j) of B-lane, (block, i, j) of D-lane
__global__ void probe(double *D, int la, int lb) {
    const int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
    D[l] = d;
}

template <int NACC, bool SMALL> void run(double *out, int waves, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(256), dim3(256 * waves), 0, 0, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(256), dim3(256 * waves), 0, 0, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * NACC * waves);
    printf("%s  waves/SIMD %d  chains %d: %.2f ns per instruction per SIMD = %.1f TFLOP/s\n", SMALL ? "4x4x4_4b " : "16x16x4  ", waves, NACC, per,
           (SMALL ? 512.0 : 2048.0) * 1024 / per * 1e-3);
}

int main() {
    double *out; hipMalloc(&out, 256 * 1024 * 8);
    const int it = 20000;
    run<1, false>(out, 1, it); run<4, false>(out, 1, it); run<4, false>(out, 2, it);
    run<1, true>(out, 1, it); run<4, true>(out, 1, it); run<8, true>(out, 1, it); run<4, true>(out, 2, it); run<8, true>(out, 2, it);
    double *D; hipMalloc(&D, 64 * 8); double h[64];
    for (int la : {0, 1, 4, 5, 16, 17, 21}) {
        printf("A one-hot at lane %2d: B lane -> D lanes:", la);
        for (int lb = 0; lb < 64; ++lb) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, D, la, lb); hipMemcpy(h, D, sizeof h, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l) if (h[l] != 0.0) printf(" %d->%d", lb, l);
        }
        printf("\n");
    }
    return 0;
}
