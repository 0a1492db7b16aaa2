// Microbenchmark (gfx950): issue cost of the vector instructions of the step kernels' epilogues, 2 waves per SIMD (the occupancy
// of the resident kernels): 16 independent instructions per loop iteration, explicit registers.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
             "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","s20","s21","s22","s23","s24","s25","vcc"

// 16 destinations v[40:41] .. v[70:71], sources v[8:9] .. v[38:39]
#define R16(OP, TAIL) \
    OP " v[40:41], v[8:9]" TAIL "\n" OP " v[42:43], v[10:11]" TAIL "\n" OP " v[44:45], v[12:13]" TAIL "\n" OP " v[46:47], v[14:15]" TAIL "\n" \
    OP " v[48:49], v[16:17]" TAIL "\n" OP " v[50:51], v[18:19]" TAIL "\n" OP " v[52:53], v[20:21]" TAIL "\n" OP " v[54:55], v[22:23]" TAIL "\n" \
    OP " v[56:57], v[24:25]" TAIL "\n" OP " v[58:59], v[26:27]" TAIL "\n" OP " v[60:61], v[28:29]" TAIL "\n" OP " v[62:63], v[30:31]" TAIL "\n" \
    OP " v[64:65], v[32:33]" TAIL "\n" OP " v[66:67], v[34:35]" TAIL "\n" OP " v[68:69], v[36:37]" TAIL "\n" OP " v[70:71], v[38:39]" TAIL "\n"
#define R16S(OP, TAIL) \
    OP " v40, v8" TAIL "\n" OP " v42, v10" TAIL "\n" OP " v44, v12" TAIL "\n" OP " v46, v14" TAIL "\n" OP " v48, v16" TAIL "\n" OP " v50, v18" TAIL "\n" \
    OP " v52, v20" TAIL "\n" OP " v54, v22" TAIL "\n" OP " v56, v24" TAIL "\n" OP " v58, v26" TAIL "\n" OP " v60, v28" TAIL "\n" OP " v62, v30" TAIL "\n" \
    OP " v64, v32" TAIL "\n" OP " v66, v34" TAIL "\n" OP " v68, v36" TAIL "\n" OP " v70, v38" TAIL "\n"

template <int V>
__global__ void kern(long long *cyc, int iters) {
    asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n s_mov_b32 s22, 3\n" ::: CLOB);
#define Z(r) asm volatile("v_mov_b32 v" #r ", 0" ::: CLOB);
    Z(8) Z(9) Z(10) Z(11) Z(12) Z(13) Z(14) Z(15) Z(16) Z(17) Z(18) Z(19) Z(20) Z(21) Z(22) Z(23) Z(24) Z(25) Z(26) Z(27) Z(28) Z(29) Z(30) Z(31)
    Z(32) Z(33) Z(34) Z(35) Z(36) Z(37) Z(38) Z(39) Z(40) Z(41) Z(42) Z(43) Z(44) Z(45) Z(46) Z(47) Z(48) Z(49) Z(50) Z(51) Z(52) Z(53) Z(54) Z(55)
    Z(56) Z(57) Z(58) Z(59) Z(60) Z(61) Z(62) Z(63) Z(64) Z(65) Z(66) Z(67) Z(68) Z(69) Z(70) Z(71)
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (V == 0) asm volatile(R16("v_mul_f64", ", s[20:21]") ::: CLOB);
        if (V == 1) asm volatile(R16("v_ldexp_f64", ", s22") ::: CLOB);
        if (V == 2) asm volatile(R16("v_rndne_f64", "") ::: CLOB);
        if (V == 3) asm volatile("v_cvt_i32_f64 v40, v[8:9]\n v_cvt_i32_f64 v42, v[10:11]\n v_cvt_i32_f64 v44, v[12:13]\n v_cvt_i32_f64 v46, v[14:15]\n"
                                 "v_cvt_i32_f64 v48, v[16:17]\n v_cvt_i32_f64 v50, v[18:19]\n v_cvt_i32_f64 v52, v[20:21]\n v_cvt_i32_f64 v54, v[22:23]\n"
                                 "v_cvt_i32_f64 v56, v[24:25]\n v_cvt_i32_f64 v58, v[26:27]\n v_cvt_i32_f64 v60, v[28:29]\n v_cvt_i32_f64 v62, v[30:31]\n"
                                 "v_cvt_i32_f64 v64, v[32:33]\n v_cvt_i32_f64 v66, v[34:35]\n v_cvt_i32_f64 v68, v[36:37]\n v_cvt_i32_f64 v70, v[38:39]\n" ::: CLOB);
        if (V == 4) asm volatile(R16S("v_add_u32", ", s22") ::: CLOB);
        if (V == 5) asm volatile(R16S("v_mov_b32", "") ::: CLOB);
        if (V == 6) asm volatile(R16S("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf") ::: CLOB);
        if (V == 7) asm volatile(R16("v_max_f64", ", s[20:21]") ::: CLOB);
        if (V == 8) asm volatile(R16("v_fma_f64", ", s[20:21], s[20:21]") ::: CLOB);
        if (V == 9) asm volatile(R16S("v_mad_u32_u24", ", s22, v9") ::: CLOB);
        if (V == 10) asm volatile(R16S("v_cndmask_b32", ", v9, vcc") ::: CLOB);
        if (V == 11) asm volatile(R16S("v_lshl_add_u32", ", 3, v9") ::: CLOB);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V> void go(long long *cyc, int iters, int threads) { hipLaunchKernelGGL(kern<V>, dim3(256), dim3(threads), 0, 0, cyc, iters); }

int main() {
    long long *cyc, hc;
    CHECK(hipMalloc(&cyc, 64));
    const int iters = 20000;
    const char *names[12] = {"v_mul_f64", "v_ldexp_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_add_u32", "v_mov_b32", "v_mov_b32_dpp", "v_max_f64", "v_fma_f64",
                             "v_mad_u32_u24", "v_cndmask_b32", "v_lshl_add_u32"};
    void (*fn[12])(long long *, int, int) = {go<0>, go<1>, go<2>, go<3>, go<4>, go<5>, go<6>, go<7>, go<8>, go<9>, go<10>, go<11>};
    for (int waves : {1, 2, 4})
        for (int v = 0; v < 12; ++v) {
            const int threads = 256 * waves;
            fn[v](cyc, iters, threads); CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0); fn[v](cyc, iters, threads); (void)hipEventRecord(e1); CHECK(hipDeviceSynchronize());
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            const double n = (double)iters * 16;
            printf("waves/SIMD %d %-14s: %.2f ns per instruction per SIMD (%.2f clock64 ticks per instruction per wave)\n", waves, names[v],
                   ms * 1e6 / (n * waves), hc / n);
        }
    return 0;
}
