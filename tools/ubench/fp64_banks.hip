// Microbenchmark (gfx950): does the fp64 issue rate depend on which VGPR banks the 64-bit source operands sit in?
// 16 independent instructions per loop iteration, explicit registers (inline asm), 1 / 2 / 4 waves per SIMD.
//   add_same : v_add_f64  d, a, b   with a, b both = 0 (mod 4)          add_diff : a = 0, b = 2 (mod 4)
//   fmac_same: v_fmac_f64 d, s, b   with d, b both = 0 (mod 4)          fmac_diff: d = 0, b = 2 (mod 4)
//   fma_1src : v_fma_f64  d, d, s, s  (one VGPR source: the pattern of fp64_rate.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
             "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
             "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103", \
             "v104","v105","v106","v107","s20","s21","s22","s23"

// sources a_k = v[8+4k .. ] (all = 0 mod 4), b_k same pair: v[12+4k], b'_k other pair: v[10+4k]; destinations v[72+2k]
#define I_ADD_SAME(k) "v_add_f64 v[" #k "*2+72:" #k "*2+73], v[8:9], v[12:13]\n"
template <int V> struct Body;
#define BODY16(name, line) \
    static __device__ __forceinline__ void name() { asm volatile(line(0) line(1) line(2) line(3) line(4) line(5) line(6) line(7) line(8) line(9) line(10) line(11) line(12) line(13) line(14) line(15) ::: CLOB); }

// (the assembler does not evaluate "k*2+72" inside register ranges: spell the 16 lines out)
#define L(d, a, b, op) op " v[" #d ":" #d "+1], " a ", " b "\n"
static __device__ __forceinline__ void add_same() {
    asm volatile(
        "v_add_f64 v[72:73], v[8:9], v[12:13]\n v_add_f64 v[74:75], v[16:17], v[20:21]\n v_add_f64 v[76:77], v[24:25], v[28:29]\n v_add_f64 v[78:79], v[32:33], v[36:37]\n"
        "v_add_f64 v[80:81], v[40:41], v[44:45]\n v_add_f64 v[82:83], v[48:49], v[52:53]\n v_add_f64 v[84:85], v[56:57], v[60:61]\n v_add_f64 v[86:87], v[64:65], v[68:69]\n"
        "v_add_f64 v[88:89], v[12:13], v[16:17]\n v_add_f64 v[90:91], v[20:21], v[24:25]\n v_add_f64 v[92:93], v[28:29], v[32:33]\n v_add_f64 v[94:95], v[36:37], v[40:41]\n"
        "v_add_f64 v[96:97], v[44:45], v[48:49]\n v_add_f64 v[98:99], v[52:53], v[56:57]\n v_add_f64 v[100:101], v[60:61], v[64:65]\n v_add_f64 v[102:103], v[68:69], v[8:9]\n" ::: CLOB);
}
static __device__ __forceinline__ void add_diff() {
    asm volatile(
        "v_add_f64 v[72:73], v[8:9], v[14:15]\n v_add_f64 v[74:75], v[16:17], v[22:23]\n v_add_f64 v[76:77], v[24:25], v[30:31]\n v_add_f64 v[78:79], v[32:33], v[38:39]\n"
        "v_add_f64 v[80:81], v[40:41], v[46:47]\n v_add_f64 v[82:83], v[48:49], v[54:55]\n v_add_f64 v[84:85], v[56:57], v[62:63]\n v_add_f64 v[86:87], v[64:65], v[70:71]\n"
        "v_add_f64 v[88:89], v[12:13], v[18:19]\n v_add_f64 v[90:91], v[20:21], v[26:27]\n v_add_f64 v[92:93], v[28:29], v[34:35]\n v_add_f64 v[94:95], v[36:37], v[42:43]\n"
        "v_add_f64 v[96:97], v[44:45], v[50:51]\n v_add_f64 v[98:99], v[52:53], v[58:59]\n v_add_f64 v[100:101], v[60:61], v[66:67]\n v_add_f64 v[102:103], v[68:69], v[10:11]\n" ::: CLOB);
}
static __device__ __forceinline__ void fmac_same() {      // dst (read as accumulator) and src1 in the same bank pair
    asm volatile(
        "v_fmac_f64 v[72:73], s[20:21], v[8:9]\n v_fmac_f64 v[76:77], s[20:21], v[16:17]\n v_fmac_f64 v[80:81], s[20:21], v[24:25]\n v_fmac_f64 v[84:85], s[20:21], v[32:33]\n"
        "v_fmac_f64 v[88:89], s[20:21], v[40:41]\n v_fmac_f64 v[92:93], s[20:21], v[48:49]\n v_fmac_f64 v[96:97], s[20:21], v[56:57]\n v_fmac_f64 v[100:101], s[20:21], v[64:65]\n"
        "v_fmac_f64 v[74:75], s[22:23], v[10:11]\n v_fmac_f64 v[78:79], s[22:23], v[18:19]\n v_fmac_f64 v[82:83], s[22:23], v[26:27]\n v_fmac_f64 v[86:87], s[22:23], v[34:35]\n"
        "v_fmac_f64 v[90:91], s[22:23], v[42:43]\n v_fmac_f64 v[94:95], s[22:23], v[50:51]\n v_fmac_f64 v[98:99], s[22:23], v[58:59]\n v_fmac_f64 v[102:103], s[22:23], v[66:67]\n" ::: CLOB);
}
static __device__ __forceinline__ void fmac_diff() {
    asm volatile(
        "v_fmac_f64 v[72:73], s[20:21], v[10:11]\n v_fmac_f64 v[76:77], s[20:21], v[18:19]\n v_fmac_f64 v[80:81], s[20:21], v[26:27]\n v_fmac_f64 v[84:85], s[20:21], v[34:35]\n"
        "v_fmac_f64 v[88:89], s[20:21], v[42:43]\n v_fmac_f64 v[92:93], s[20:21], v[50:51]\n v_fmac_f64 v[96:97], s[20:21], v[58:59]\n v_fmac_f64 v[100:101], s[20:21], v[66:67]\n"
        "v_fmac_f64 v[74:75], s[22:23], v[8:9]\n v_fmac_f64 v[78:79], s[22:23], v[16:17]\n v_fmac_f64 v[82:83], s[22:23], v[24:25]\n v_fmac_f64 v[86:87], s[22:23], v[32:33]\n"
        "v_fmac_f64 v[90:91], s[22:23], v[40:41]\n v_fmac_f64 v[94:95], s[22:23], v[48:49]\n v_fmac_f64 v[98:99], s[22:23], v[56:57]\n v_fmac_f64 v[102:103], s[22:23], v[64:65]\n" ::: CLOB);
}
static __device__ __forceinline__ void fma_1src() {
    asm volatile(
        "v_fma_f64 v[72:73], v[72:73], s[20:21], s[20:21]\n v_fma_f64 v[74:75], v[74:75], s[20:21], s[20:21]\n v_fma_f64 v[76:77], v[76:77], s[20:21], s[20:21]\n v_fma_f64 v[78:79], v[78:79], s[20:21], s[20:21]\n"
        "v_fma_f64 v[80:81], v[80:81], s[20:21], s[20:21]\n v_fma_f64 v[82:83], v[82:83], s[20:21], s[20:21]\n v_fma_f64 v[84:85], v[84:85], s[20:21], s[20:21]\n v_fma_f64 v[86:87], v[86:87], s[20:21], s[20:21]\n"
        "v_fma_f64 v[88:89], v[88:89], s[20:21], s[20:21]\n v_fma_f64 v[90:91], v[90:91], s[20:21], s[20:21]\n v_fma_f64 v[92:93], v[92:93], s[20:21], s[20:21]\n v_fma_f64 v[94:95], v[94:95], s[20:21], s[20:21]\n"
        "v_fma_f64 v[96:97], v[96:97], s[20:21], s[20:21]\n v_fma_f64 v[98:99], v[98:99], s[20:21], s[20:21]\n v_fma_f64 v[100:101], v[100:101], s[20:21], s[20:21]\n v_fma_f64 v[102:103], v[102:103], s[20:21], s[20:21]\n" ::: CLOB);
}
// the stencil's real mix: 8 adds (2 VGPR sources) then 8 fmacs (accumulator + the sum), banks as the compiler happened to pick = mixed
static __device__ __forceinline__ void mix_alt() {         // adds with different-bank sources, fmacs with acc / src in different banks
    asm volatile(
        "v_add_f64 v[72:73], v[8:9], v[14:15]\n v_add_f64 v[76:77], v[16:17], v[22:23]\n v_add_f64 v[80:81], v[24:25], v[30:31]\n v_add_f64 v[84:85], v[32:33], v[38:39]\n"
        "v_add_f64 v[88:89], v[40:41], v[46:47]\n v_add_f64 v[92:93], v[48:49], v[54:55]\n v_add_f64 v[96:97], v[56:57], v[62:63]\n v_add_f64 v[100:101], v[64:65], v[70:71]\n"
        "v_fmac_f64 v[74:75], s[20:21], v[72:73]\n v_fmac_f64 v[78:79], s[20:21], v[76:77]\n v_fmac_f64 v[82:83], s[20:21], v[80:81]\n v_fmac_f64 v[86:87], s[20:21], v[84:85]\n"
        "v_fmac_f64 v[90:91], s[20:21], v[88:89]\n v_fmac_f64 v[94:95], s[20:21], v[92:93]\n v_fmac_f64 v[98:99], s[20:21], v[96:97]\n v_fmac_f64 v[102:103], s[20:21], v[100:101]\n" ::: CLOB);
}

template <int V>
__global__ void kern(long long *cyc, int iters) {
    asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0x3fe00000\n" ::: CLOB);
#define Z(r) asm volatile("v_mov_b32 v" #r ", 0" ::: CLOB);
    Z(8) Z(9) Z(10) Z(11) Z(12) Z(13) Z(14) Z(15) Z(16) Z(17) Z(18) Z(19) Z(20) Z(21) Z(22) Z(23) Z(24) Z(25) Z(26) Z(27) Z(28) Z(29) Z(30) Z(31)
    Z(32) Z(33) Z(34) Z(35) Z(36) Z(37) Z(38) Z(39) Z(40) Z(41) Z(42) Z(43) Z(44) Z(45) Z(46) Z(47) Z(48) Z(49) Z(50) Z(51) Z(52) Z(53) Z(54) Z(55)
    Z(56) Z(57) Z(58) Z(59) Z(60) Z(61) Z(62) Z(63) Z(64) Z(65) Z(66) Z(67) Z(68) Z(69) Z(70) Z(71) Z(72) Z(73) Z(74) Z(75) Z(76) Z(77) Z(78) Z(79)
    Z(80) Z(81) Z(82) Z(83) Z(84) Z(85) Z(86) Z(87) Z(88) Z(89) Z(90) Z(91) Z(92) Z(93) Z(94) Z(95) Z(96) Z(97) Z(98) Z(99) Z(100) Z(101) Z(102) Z(103)
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (V == 0) add_same();
        if (V == 1) add_diff();
        if (V == 2) fmac_same();
        if (V == 3) fmac_diff();
        if (V == 4) fma_1src();
        if (V == 5) mix_alt();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    long long *cyc, hc;
    CHECK(hipMalloc(&cyc, 64));
    const int iters = 20000;
    const char *names[6] = {"add_same", "add_diff", "fmac_same", "fmac_diff", "fma_1src", "mix_alt"};
    for (int waves : {1, 2, 4}) {
        for (int v = 0; v < 6; ++v) {
            const int threads = 256 * waves;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto go = [&]() {
                switch (v) {
                    case 0: hipLaunchKernelGGL(kern<0>, dim3(256), dim3(threads), 0, 0, cyc, iters); break;
                    case 1: hipLaunchKernelGGL(kern<1>, dim3(256), dim3(threads), 0, 0, cyc, iters); break;
                    case 2: hipLaunchKernelGGL(kern<2>, dim3(256), dim3(threads), 0, 0, cyc, iters); break;
                    case 3: hipLaunchKernelGGL(kern<3>, dim3(256), dim3(threads), 0, 0, cyc, iters); break;
                    case 4: hipLaunchKernelGGL(kern<4>, dim3(256), dim3(threads), 0, 0, cyc, iters); break;
                    default: hipLaunchKernelGGL(kern<5>, dim3(256), dim3(threads), 0, 0, cyc, iters); break;
                }
            };
            go(); hipDeviceSynchronize();
            hipEventRecord(e0); go(); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            const double n = (double)iters * 16;
            printf("waves/SIMD %d %-10s: %.2f cycles per instruction per wave = %.2f per SIMD; kernel %.3f ms = %.2f G wave-instr/s/SIMD\n", waves, names[v],
                   hc / n, hc / n / waves, ms, n * waves * 1024.0 / (ms * 1e-3) / 1e9 / 1024.0);
        }
    }
    return 0;
}
