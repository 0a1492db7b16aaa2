// v_permlane16_swap on gfx950: which rows of 16 lanes change places?  hipcc --offload-arch=gfx950 -O3 permlane16.hip -o permlane16 && ./permlane16
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row) std::printf("row %d: first operand now holds %u.., second operand now holds %u..\n", row, h[16 * row], h[64 + 16 * row]);
    return 0;
}
