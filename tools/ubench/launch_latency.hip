// Launch-to-launch latency of small DEPENDENT kernels on one stream: plain launches vs one hipGraph of the same chain (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o launch_latency launch_latency.hip && ./launch_latency
// Each kernel: `blocks` blocks x 256 threads, reads a value the previous kernel wrote (a real dependency), ~`work` FMAs per thread.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void step(const double *in, double *out, int work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = in[i];
    for (int k = 0; k < work; ++k) v = fma(v, 0.999999, 1e-9);
    out[i] = v;
}
int main() {
    const int N = 2000;
    for (int blocks : {16, 256}) for (int work : {16, 2048}) {
        double *a, *b;
        CK(hipMalloc(&a, blocks * 256 * 8)); CK(hipMalloc(&b, blocks * 256 * 8));
        CK(hipMemset(a, 0, blocks * 256 * 8));
        hipStream_t s; CK(hipStreamCreate(&s));
        auto run = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step, dim3(blocks), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, work); };
        run(); CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        run(); CK(hipStreamSynchronize(s));
        const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        run();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("blocks %4d work %5d: plain %.2f us per kernel, graph %.2f us per kernel\n", blocks, work, plain, graph);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(a)); CK(hipFree(b)); CK(hipStreamDestroy(s));
    }
    return 0;
}
