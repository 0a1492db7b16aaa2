/* Plain-C use of libblhip.so (include/blhip.h): the coal-mining example of the reference's first tutorial
 * (Poisson rate, Gaussian random walk, 200-point grid; SURVEY.md 8(d) configuration C1) without any Python.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -o c_abi_demo -Lbayesloop_amd -lblhip -Wl,-rpath,$PWD/bayesloop_amd -lm
 *   ./c_abi_demo            (needs an MI355X; prints the ABI version and exits with 0 when no device is visible)
 *
 * Expected log-evidence: -171.68672187433867 (reference value).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "blhip.h"

int main(void) {
    static const double coal[110] = {5, 4, 1, 0, 4, 3, 4, 0, 6, 3, 3, 4, 0, 2, 6, 3, 3, 5, 4, 5, 3, 1, 4, 4, 1, 5, 5, 3, 4, 2, 5, 2, 2, 3, 4, 2,
                                     1, 3, 2, 2, 1, 1, 1, 1, 3, 0, 0, 1, 0, 1, 1, 0, 0, 3, 1, 0, 3, 2, 2, 0, 1, 1, 1, 0, 1, 0, 1, 0, 0, 0, 2, 1,
                                     0, 0, 0, 1, 1, 0, 2, 3, 3, 1, 1, 2, 1, 1, 1, 1, 2, 3, 3, 0, 0, 0, 1, 4, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0,
                                     1, 0};
    enum { T = 110, N = 200 };
    double grid[N], prior[N], ts[T], local[T], means[T], logE = 0.0, sum = 0.0;
    int64_t abort_step = -1;
    int32_t abort_phase = 0;
    int k;

    printf("libblhip ABI version %d (header %d), %d HIP device(s)\n", blhip_abi_version(), BLHIP_ABI_VERSION, blhip_device_count());
    if (blhip_abi_version() != BLHIP_ABI_VERSION) return 2;
    if (blhip_device_count() <= 0) return 0;

    for (k = 0; k < N; ++k) grid[k] = 6.0 * (k + 1) / (N + 1);            /* bl.oint(0, 6, 200), helper.py:90-104 */
    for (k = 0; k < N; ++k) { prior[k] = sqrt(1.0 / grid[k]); sum += prior[k]; }   /* Jeffreys prior, observationModels.py:486-487 */
    for (k = 0; k < N; ++k) prior[k] = prior[k] / sum / (grid[1] - grid[0]);       /* core.py:224-235 */
    for (k = 0; k < T; ++k) ts[k] = 1852 + k;

    blhip_op op = {BLHIP_OP_GRW, 0, -1, 0};
    double sigma = 0.2;
    blhip_problem p;
    blhip_result r;
    blhip_ctx *ctx = blhip_create(0);
    if (!ctx) { fprintf(stderr, "blhip_create: %s\n", blhip_last_error(NULL)); return 1; }

    p.ndim = 1; p.obs_model = BLHIP_OM_POISSON;
    p.n[0] = N; p.n[1] = 1; p.marginal[0] = grid; p.marginal[1] = NULL;
    p.lattice[0] = grid[1] - grid[0]; p.lattice[1] = 1.0;
    p.T = T; p.seg_len = 1; p.data_dim = 1; p.data = coal; p.timestamps = ts;
    p.prior = prior; p.reset_prior = NULL; p.indep_prior = NULL; p.lik = NULL;
    p.n_ops = 1; p.ops = &op; p.resume_time = -1.0; p.carry_slot = 0; p.reserved0 = 0; p.backward_init = NULL; p.prior_token = 0;
    r.log_evidence = &logE; r.local_evidence = local; r.posterior_mean = means;
    r.abort_step = &abort_step; r.abort_phase = &abort_phase;

    if (blhip_fit(ctx, &p, 1, &sigma, NULL, 0u, &r) != 0) { fprintf(stderr, "blhip_fit: %s\n", blhip_last_error(ctx)); return 1; }
    printf("log-evidence %.12f (reference -171.686721874339), mean rate 1852: %.4f, 1961: %.4f\n", logE, means[0], means[T - 1]);
    blhip_destroy(ctx);
    return fabs(logE - (-171.68672187433867)) < 1e-7 ? 0 : 3;
}
