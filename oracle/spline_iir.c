/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/bl_oracle.py).
 *
 * scipy.ndimage's cubic-spline prefilter for mode='nearest' / 'reflect', restated: the one third-party recursion behind
 * bayesloop/transitionModels.py:581 and :600 (Deterministic: scipy.ndimage.shift(order=3, mode='nearest')).  SciPy (requirement
 * scipy>=0.17.1 in the reference's setup.py:14, unpinned; 1.15.3 in this image) is not under /root/reference; this follows its
 * published algorithm (scipy/ndimage/src/ni_splines.c: filter_gain, apply_filter, _init_causal_reflect, _init_anticausal_reflect):
 *
 *     gain = (1 - z)(1 - 1/z);  c *= gain
 *     c[0] <- c0 + z / (1 - z^2n) * sum_{i=0..n-1} z^i (c[i] + z^n c[n-1-i])      (causal initialisation, whole row)
 *     c[i] += z c[i-1]                       i = 1 .. n-1                           (causal pass)
 *     c[n-1] *= z / (z - 1)                                                         (anti-causal initialisation)
 *     c[i] = z (c[i+1] - c[i])               i = n-2 .. 0                           (anti-causal pass)
 *
 * with the pole z = -0.2679491924311227 (SciPy's decimal literal of sqrt(3) - 2; sqrt(3.) - 2. in double arithmetic is TWO ulp away and
 * does not reproduce SciPy's bits).  Compiled with -ffp-contract=off: no fused multiply-add, the operation order above, so the result is
 * bit-identical to scipy.ndimage.spline_filter1d(x, 3, mode='nearest') (tests/test_filter_restatement.py checks exactly that).
 */
#include <math.h>

#define BL_SPLINE_POLE (-0.2679491924311227)

/* rows: `count` rows of `n` doubles each, row r at c + r * row_stride, elements contiguous */
void bl_spline_prefilter_reflect(double *c0, long n, long count, long row_stride)
{
    const double z = BL_SPLINE_POLE;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    const double z_n = pow(z, (double) n);
    for (long r = 0; r < count; ++r) {
        double *c = c0 + r * row_stride;
        long i;
        for (i = 0; i < n; ++i)
            c[i] *= gain;
        if (n < 2)
            continue;
        {
            double z_i = z;
            const double first = c[0];
            c[0] = c[0] + z_n * c[n - 1];
            for (i = 1; i < n; ++i) {
                c[0] += z_i * (c[i] + z_n * c[n - 1 - i]);
                z_i *= z;
            }
            c[0] *= z / (1.0 - z_n * z_n);
            c[0] += first;
        }
        for (i = 1; i < n; ++i)
            c[i] += z * c[i - 1];
        c[n - 1] *= z / (z - 1.0);
        for (i = n - 2; i >= 0; --i)
            c[i] = z * (c[i + 1] - c[i]);
    }
}
