// Microbenchmark (gfx950): what a v_cndmask_b32 costs next to the fp64 instructions of the backward epilogues (the per-cell select of
// p / L's 0 / 0 -> NaN rule, core.py:463).  tools/ubench/valu_rates.hip measured 9.8 ns per instruction and SIMD at ANY occupancy for the
// VOP2 form with vcc -- 3.4 x a v_mul_f64 at two waves per SIMD; here: the VOP3 form with a scalar-register mask, both forms interleaved
// with multiplications, and the compare + two selects of a 64-bit value as the compiler emits them.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cndmask_rate.hip -o /tmp/cndmask_rate && /tmp/cndmask_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
             "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","s20","s21","s22","s23","s24","s25","vcc"
#define M(d, s) "v_mul_f64 v[" #d "], v[" #s "], s[20:21]\n"
#define C32(d, s) "v_cndmask_b32 v" #d ", v" #s ", v9, vcc\n"
#define C64(d, s) "v_cndmask_b32 v" #d ", v" #s ", v9, s[24:25]\n"

template <int V>
__global__ void kern(long long *cyc, int iters) {
    asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n s_mov_b32 s22, 3\n s_mov_b64 s[24:25], 0x5555\n s_mov_b64 vcc, 0x3333\n" ::: CLOB);
#define Z(r) asm volatile("v_mov_b32 v" #r ", 0" ::: CLOB);
    Z(8) Z(9) Z(10) Z(11) Z(12) Z(13) Z(14) Z(15) Z(16) Z(17) Z(18) Z(19) Z(20) Z(21) Z(22) Z(23) Z(24) Z(25) Z(26) Z(27) Z(28) Z(29) Z(30) Z(31)
    Z(32) Z(33) Z(34) Z(35) Z(36) Z(37) Z(38) Z(39) Z(40) Z(41) Z(42) Z(43) Z(44) Z(45) Z(46) Z(47) Z(48) Z(49) Z(50) Z(51) Z(52) Z(53) Z(54) Z(55)
    Z(56) Z(57) Z(58) Z(59) Z(60) Z(61) Z(62) Z(63) Z(64) Z(65) Z(66) Z(67) Z(68) Z(69) Z(70) Z(71)
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (V == 0) asm volatile(M(40:41, 8:9) M(42:43, 10:11) M(44:45, 12:13) M(46:47, 14:15) M(48:49, 16:17) M(50:51, 18:19) M(52:53, 20:21) M(54:55, 22:23)
                                 M(56:57, 24:25) M(58:59, 26:27) M(60:61, 28:29) M(62:63, 30:31) M(64:65, 32:33) M(66:67, 34:35) M(68:69, 36:37) M(70:71, 38:39) ::: CLOB);
        if (V == 1) asm volatile(C32(40, 8) C32(42, 10) C32(44, 12) C32(46, 14) C32(48, 16) C32(50, 18) C32(52, 20) C32(54, 22)
                                 C32(56, 24) C32(58, 26) C32(60, 28) C32(62, 30) C32(64, 32) C32(66, 34) C32(68, 36) C32(70, 38) ::: CLOB);
        if (V == 2) asm volatile(C64(40, 8) C64(42, 10) C64(44, 12) C64(46, 14) C64(48, 16) C64(50, 18) C64(52, 20) C64(54, 22)
                                 C64(56, 24) C64(58, 26) C64(60, 28) C64(62, 30) C64(64, 32) C64(66, 34) C64(68, 36) C64(70, 38) ::: CLOB);
        // 12 multiplications + 4 selects (the backward epilogue's mix per two cells)
        if (V == 3) asm volatile(M(40:41, 8:9) M(42:43, 10:11) M(44:45, 12:13) C32(46, 14) M(48:49, 16:17) M(50:51, 18:19) M(52:53, 20:21) C32(54, 22)
                                 M(56:57, 24:25) M(58:59, 26:27) M(60:61, 28:29) C32(62, 30) M(64:65, 32:33) M(66:67, 34:35) M(68:69, 36:37) C32(70, 38) ::: CLOB);
        if (V == 4) asm volatile(M(40:41, 8:9) M(42:43, 10:11) M(44:45, 12:13) C64(46, 14) M(48:49, 16:17) M(50:51, 18:19) M(52:53, 20:21) C64(54, 22)
                                 M(56:57, 24:25) M(58:59, 26:27) M(60:61, 28:29) C64(62, 30) M(64:65, 32:33) M(66:67, 34:35) M(68:69, 36:37) C64(70, 38) ::: CLOB);
        // compare + two selects (a 64-bit value), four times, + 4 multiplications: 16 instructions
        if (V == 5) asm volatile("v_cmp_neq_f64 vcc, 0, v[8:9]\n" C32(40, 10) C32(41, 11) M(42:43, 12:13)
                                 "v_cmp_neq_f64 vcc, 0, v[14:15]\n" C32(44, 16) C32(45, 17) M(46:47, 18:19)
                                 "v_cmp_neq_f64 vcc, 0, v[20:21]\n" C32(48, 22) C32(49, 23) M(50:51, 24:25)
                                 "v_cmp_neq_f64 vcc, 0, v[26:27]\n" C32(52, 28) C32(53, 29) M(54:55, 30:31) ::: CLOB);
        // ... the same compares accumulated in a scalar mask instead (no select per cell): 4 compares + 4 scalar ORs + 8 multiplications + 4 more
        if (V == 6) asm volatile("v_cmp_eq_f64 vcc, 0, v[8:9]\n s_or_b64 s[24:25], s[24:25], vcc\n" M(40:41, 10:11) M(42:43, 12:13)
                                 "v_cmp_eq_f64 vcc, 0, v[14:15]\n s_or_b64 s[24:25], s[24:25], vcc\n" M(44:45, 16:17) M(46:47, 18:19)
                                 "v_cmp_eq_f64 vcc, 0, v[20:21]\n s_or_b64 s[24:25], s[24:25], vcc\n" M(48:49, 22:23) M(50:51, 24:25)
                                 "v_cmp_eq_f64 vcc, 0, v[26:27]\n s_or_b64 s[24:25], s[24:25], vcc\n" M(52:53, 28:29) M(54:55, 30:31)
                                 M(56:57, 32:33) M(58:59, 34:35) M(60:61, 36:37) M(62:63, 38:39) ::: CLOB);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V> void go(long long *cyc, int iters, int threads) { hipLaunchKernelGGL(kern<V>, dim3(256), dim3(threads), 0, 0, cyc, iters); }

int main() {
    long long *cyc;
    CHECK(hipMalloc(&cyc, 64));
    const int iters = 20000;
    const char *names[7] = {"16 v_mul_f64", "16 v_cndmask_b32 (vcc)", "16 v_cndmask_b32 (sgpr pair)", "12 mul + 4 cndmask (vcc)", "12 mul + 4 cndmask (sgpr)",
                            "4 x (cmp + 2 cndmask + mul)", "4 x (cmp + s_or + 2 mul) + 4 mul"};
    void (*fn[7])(long long *, int, int) = {go<0>, go<1>, go<2>, go<3>, go<4>, go<5>, go<6>};
    for (int waves : {2, 4})
        for (int v = 0; v < 7; ++v) {
            const int threads = 256 * waves;
            fn[v](cyc, iters, threads); CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0); fn[v](cyc, iters, threads); (void)hipEventRecord(e1); CHECK(hipDeviceSynchronize());
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d %-36s: %.2f ns per 16-instruction group per SIMD\n", waves, names[v], ms * 1e6 / ((double)iters * waves));
        }
    return 0;
}
