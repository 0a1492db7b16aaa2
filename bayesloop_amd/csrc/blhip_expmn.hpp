// exp() as mantissa * 2^exponent: the building block of the Gaussian likelihood recurrence of the step kernels.
// Compiles for the device (hipcc) and for the host (g++: the CPU emulation of the time-resident kernel's index logic in
// tools/emu/, a development check -- never a product path).
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BL_HD __host__ __device__ __forceinline__
#else
#define BL_HD inline
#endif

namespace blmath {

// exp(a) = m * 2^n with m in [0.70, 1.42]; never overflows / underflows.  |error| < 2e-16 relative.
BL_HD void exp_mn(double a, double &m, int &n) {
    a = fmin(fmax(a, -1.4e9), 1.4e9);
    const double kn = rint(a * 1.44269504088896340736);
    double r = fma(-kn, 6.93147180369123816490e-01, a);
    r = fma(-kn, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = fma(p, r, 2.0876756987868100e-09);        // 1/12!
    p = fma(p, r, 2.5052108385441720e-08);        // 1/11!
    p = fma(p, r, 2.7557319223985893e-07);        // 1/10!
    p = fma(p, r, 2.7557319223985888e-06);        // 1/9!
    p = fma(p, r, 2.4801587301587302e-05);        // 1/8!
    p = fma(p, r, 1.9841269841269841e-04);        // 1/7!
    p = fma(p, r, 1.3888888888888889e-03);        // 1/6!
    p = fma(p, r, 8.3333333333333332e-03);        // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);        // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    m = p;
    n = (int)kn;
}

// 1 / m for the mantissa of exp_mn: exp(-a) = (1 / m) * 2^(-n) with the SAME n (rint is odd), so the reciprocal recurrence of the backward
// kernels (p / L without a division per cell) needs no second exponential per anchor -- v_rcp_f64 + one Newton step (3 instructions; an
// exp_mn is ~35 with its constants).  |error| < 2e-16 relative.
BL_HD double inv_m(double m) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(m);
    return fma(fma(-m, r, 1.0), r, r);
#else
    return 1.0 / m;
#endif
}

}  // namespace blmath
