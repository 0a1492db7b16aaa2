/*
 * blhip.h -- C-ABI of libblhip.so: the MI355X (gfx950) implementation of bayesloop's grid-based
 * forward-backward inference loop.
 *
 * The reference (christophmark/bayesloop v1.5.7) is pure Python and has NO FFI for this path; its boundary is the
 * duck-typed method level.  Each entry point below names the reference interface it stands in for
 * (file:line relative to the reference checkout):
 *
 *   blhip_fit            Study.fit                        bayesloop/core.py:330-486
 *                        (+ the per-hyper-point loop of   bayesloop/core.py:1349-1366 / 1473-1489 when n_chains > 1)
 *   blhip_accum_*        the evidence-weighted average    bayesloop/core.py:1295, 1362-1366, 1339, 1375-1382, 1416-1419
 *   blhip_posterior_*    the posteriorSequence attribute  bayesloop/core.py:356, 408, 436-441
 *   blhip_carry_*        OnlineStudy.step                 bayesloop/core.py:2062-2226 (with BLHIP_RESUME / BLHIP_CARRY fits)
 *   blhip_comm_*         HyperStudy.fit(nJobs > 1):       bayesloop/core.py:1307-1340 (pool.map fan-out and the merge of the
 *                        sub-studies), _parallelFit       bayesloop/core.py:1443-1495 -- one process per GPU, RCCL over xGMI
 *
 * inside which the library evaluates
 *   ObservationModel.processedPdf + Poisson/Gaussian/GaussianMean.pdf   observationModels.py:35-56, 502, 566-567, 705-706
 *   Bernoulli / Laplace / WhiteNoise / AR1 / ScaledAR1.pdf               observationModels.py:428-430, 635, 767, 830-831, 893-896
 *   GaussianRandomWalk / CombinedTransitionModel / ChangePoint / Static  transitionModels.py:49-63, 96-118, 289-317, 632-662
 *   RegimeSwitch / Independent / SerialTransitionModel + BreakPoint      transitionModels.py:339-363, 394-415, 756-818
 *   NotEqual / AlphaStable- / BivariateRandomWalk / Deterministic        transitionModels.py:450-474, 158-260, 872-911, 548-606
 *   (scipy.ndimage.gaussian_filter1d, mode='reflect', truncate=4.0, called at transitionModels.py:111;
 *    scipy.signal.fftconvolve / convolve2d at :258, :889; scipy.ndimage.shift(order=3, mode='nearest') at :581, :600).
 *
 * Conventions
 *   - plain C, no C++ / torch types; every array is float64 (double) or the integer type shown, C-contiguous.
 *   - the caller owns every host buffer and keeps it alive for the duration of the call; the library owns all
 *     device memory inside the opaque context.  No callbacks.  Nothing throws across the ABI.
 *   - return value: 0 = ok, < 0 = hard error (message via blhip_last_error).  Numerical failure of a chain (a zero
 *     normaliser, core.py:390-400 / 442-452) is NOT an error: it is reported per chain in abort_step/abort_phase and
 *     the caller mirrors the reference (logEvidence = -inf, early return).
 *   - one context per device; calls on one context must be serialised by the caller.
 */
#ifndef BLHIP_H
#define BLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLHIP_ABI_VERSION 8

typedef struct blhip_ctx blhip_ctx;

/* observation models (likelihood evaluated on the device from the data point and the grid) */
enum {
    BLHIP_OM_POISSON       = 1,   /* observationModels.py:502      1 parameter  (rate)            */
    BLHIP_OM_GAUSSIAN      = 2,   /* observationModels.py:566-567  2 parameters (mean, std)       */
    BLHIP_OM_GAUSSIAN_MEAN = 3,   /* observationModels.py:705-706  1 parameter  (mean); data (T,1,2) = (value, std) */
    /* closed-form models whose likelihood table (T,G) is built ON THE DEVICE from the data (no host pdf, no upload): */
    BLHIP_OM_BERNOULLI     = 4,   /* observationModels.py:419-439  1 parameter  (p)                               */
    BLHIP_OM_LAPLACE       = 5,   /* observationModels.py:624-635  2 parameters (mean, scale)                     */
    BLHIP_OM_WHITE_NOISE   = 6,   /* observationModels.py:756-767  1 parameter  (sigma)                           */
    BLHIP_OM_AR1           = 7,   /* observationModels.py:819-831  2 parameters (rho, sigma); seg_len 2           */
    BLHIP_OM_SCALED_AR1    = 8,   /* observationModels.py:881-896  2 parameters (rho, sigma); seg_len 2           */
    BLHIP_OM_TABLE         = 100  /* likelihood evaluated by the caller (any ObservationModel.pdf): lik (T,G) */
};

/* transition-model ops, applied in list order in both directions (transitionModels.py:645-649, 656-660) */
enum {
    BLHIP_OP_STATIC       = 0,    /* transitionModels.py:49-63    no hyper-parameter                           */
    BLHIP_OP_GRW          = 1,    /* transitionModels.py:96-118   value = sigma, axis = target parameter index  */
    BLHIP_OP_CHANGEPOINT  = 2,    /* transitionModels.py:289-317  value = tChange (flags bit 0: see below)      */
    BLHIP_OP_REGIMESWITCH = 3,    /* transitionModels.py:394-415  value = log10 pMin: clamp from below, renormalise */
    BLHIP_OP_INDEPENDENT  = 4,    /* transitionModels.py:339-363  restart from the normalised prior at every step */
    BLHIP_OP_BREAKPOINT   = 5,    /* transitionModels.py:821-840  value = tBreak: boundary between two sub-models of a
                                     SerialTransitionModel (transitionModels.py:756-786) */
    BLHIP_OP_NOTEQUAL     = 6,    /* transitionModels.py:450-474  value = log10 pMin: max(p) - p, renormalise, clamp from
                                     below, renormalise */
    BLHIP_OP_BIVARIATE    = 7,    /* transitionModels.py:872-911  value = sigma1; MUST be followed by two BIVARIATE_ARG ops
                                     carrying sigma2 and rho: dense 2-D convolution with the bivariate normal kernel on
                                     |x| <= 3 ceil(sigma / lattice), zero boundary, renormalised (2-D grids only) */
    BLHIP_OP_BIVARIATE_ARG = 8,   /* value = sigma2 (first) / rho (second) of the BIVARIATE op before it */
    BLHIP_OP_ALPHASTABLE  = 9,    /* transitionModels.py:158-260  value = scale c, axis = target parameter; MUST be followed
                                     by one ALPHASTABLE_ARG op carrying alpha: convolution along the axis with the symmetric
                                     alpha-stable density (inverse FFT of exp(-|c w|^alpha)), zero boundary, renormalised */
    BLHIP_OP_ALPHASTABLE_ARG = 10,/* value = alpha of the ALPHASTABLE op before it */
    BLHIP_OP_DETERMINISTIC = 11,  /* transitionModels.py:548-606  axis = target parameter; no value of its own; MUST be
                                     followed by 2 T DETERMINISTIC_ARG ops whose values are the shifts (in parameter units)
                                     the caller evaluated from the model's function: first T: f(t'+1) - f(t') of the
                                     transition INTO forward step i (t' = time stamp of step i-1; entry 0 is used with
                                     BLHIP_RESUME only), next T: f(t'-1) - f(t') of the backward transition into step i
                                     (t' = time stamp of step i+1; entry T-1 unused).  Cubic-spline shift as
                                     scipy.ndimage.shift(order=3, mode='nearest'), renormalised; |shift| <= 12 grid cells */
    BLHIP_OP_DETERMINISTIC_ARG = 12
};

/* A SerialTransitionModel (transitionModels.py:665-818) is flattened into the same program: the ops of its n sub-models
 * carry segment = 0..n-1, its n-1 boundaries are BREAKPOINT ops or CHANGEPOINT ops with flags bit 0 set (in list order).
 * At time stamp t the active segment is the number of boundary values <= t (:768); ops with segment >= 0 act only
 * while their segment is active; a boundary change-point additionally restarts from the prior at t == value (:801-813). */
typedef struct {
    int32_t kind;                 /* BLHIP_OP_* */
    int32_t axis;                 /* GRW: index of the target parameter (0 .. ndim-1) */
    int32_t segment;              /* -1: always active; >= 0: sub-model index of the serial model */
    int32_t flags;                /* CHANGEPOINT: bit 0 = boundary of the serial model */
} blhip_op;

#define BLHIP_MAX_DIM 4

/* The fit problem: grid, data, prior, transition program (shared by all chains of a call). */
typedef struct {
    int32_t        ndim;          /* number of observation-model parameters = grid dimensions (core.py:130-176 builds a
                                     meshgrid over any number): 1 .. BLHIP_MAX_DIM.  3 and more: BLHIP_OM_TABLE models
                                     (the reference's SciPy / SymPy / NumPy plug-ins) with GRW / STATIC / CHANGEPOINT ops */
    int32_t        obs_model;     /* BLHIP_OM_*                                                                  */
    int64_t        n[BLHIP_MAX_DIM];          /* grid size per parameter (core.py:157)                           */
    const double  *marginal[BLHIP_MAX_DIM];   /* marginal grid values per parameter, n[k] doubles (core.py:156)  */
    double         lattice[BLHIP_MAX_DIM];    /* lattice constants (core.py:161-166)                             */
    int64_t        T;             /* number of formatted time steps                                              */
    int32_t        seg_len;       /* segment length of the observation model (1 for the device-side models)      */
    int32_t        data_dim;      /* trailing data dimension d (1 for scalar series); GAUSSIAN_MEAN: 2           */
    const double  *data;          /* formatted data (T, seg_len, data_dim), NaN = missing (observationModels.py:53) */
    const double  *timestamps;    /* formatted timestamps (T,) (core.py:350)                                     */
    const double  *prior;         /* alpha_0: Study._computePrior() on the grid, (G,) (core.py:363)              */
    const double  *reset_prior;   /* what a change-point resets to, (G,) (transitionModels.py:300-312); NULL if no CHANGEPOINT op */
    const double  *indep_prior;   /* what INDEPENDENT restarts from, (G,) (transitionModels.py:351-360); NULL if no such op */
    const double  *lik;           /* BLHIP_OM_TABLE only: likelihood (T, G) evaluated by the caller; else NULL   */
    int32_t        n_ops;         /* length of the transition program                                            */
    const blhip_op *ops;
    /* streaming use (OnlineStudy.step, core.py:2062-2226): see BLHIP_RESUME / BLHIP_CARRY */
    double         resume_time;   /* BLHIP_RESUME: the time stamp the transition into step 0 is evaluated at
                                     (OnlineStudy passes len(formattedData) - 1 = -1, core.py:2164-2165)         */
    int32_t        carry_slot;    /* which carried state (one per transition model of an OnlineStudy), >= 0      */
    int32_t        reserved0;
    /* (ABI v7) the backward message entering the LAST time step, (G,); NULL: uniform 1 / G (core.py:424-425).  Lets a caller that
     * applies the transition model ITSELF -- the reference's plug-in boundary TransitionModel.computeBackwardPrior(posterior, t),
     * transitionModels.py:49-63, for models the library has no op for -- run the backward recursion one step per call: the
     * likelihood products, normalisations, sums and means of core.py:434-470 stay on the device (bayesloop_amd/core.py:
     * Study._fitHostTransition).  Fits with a backward_init take the launch-per-step kernels. */
    const double  *backward_init;
    /* (ABI v8) 0, or the caller's name for the CONTENT of `prior`: a caller that passes the same non-zero token again promises that the
     * array holds the same values as when it last passed that token.  The library then skips the upload when the context still holds
     * that prior (same token, same grid, nothing uploaded over it since) -- the prior of a 2048 x 2048 grid is 32 MiB, 0.6 ms of PCIe per
     * fit of a study that is fitted again and again (Study.optimize, core.py:488-565; repeated fit() calls).  bayesloop_amd passes a
     * token for the read-only prior arrays it caches per study (bayesloop_amd/core.py: _computePrior). */
    uint64_t       prior_token;
} blhip_problem;

/* Flags of blhip_fit */
#define BLHIP_FORWARD_ONLY   1u   /* Study.fit(forwardOnly=True)   core.py:422                                   */
#define BLHIP_EVIDENCE_ONLY  2u   /* Study.fit(evidenceOnly=True)  core.py:355, 407, 422, 477                    */
#define BLHIP_KEEP_POSTERIOR 4u   /* keep each chain's normalised posterior sequence on the device (blhip_posterior_read) */
#define BLHIP_ACCUMULATE     8u   /* fold each finite chain into the context's average-posterior accumulator (blhip_accum_*) */
#define BLHIP_RESUME         16u  /* step 0 consumes T_fwd(carried state of carry_slot, resume_time) of every chain instead of
                                     the prior (core.py:2164-2165); the slot must hold n_chains states of this grid  */
#define BLHIP_CARRY          32u  /* keep every chain's filtered, normalised distribution of the LAST step in carry_slot
                                     (core.py:2173 parameterPosterior[i][j]); forward-only / evidence-only fits      */

/* Per-chain results; every pointer may be NULL. */
typedef struct {
    double  *log_evidence;        /* (n_chains,)          Study.logEvidence (core.py:403, 417); -inf on abort    */
    double  *local_evidence;      /* (n_chains, T)        Study.localEvidence (core.py:404, 463-464)             */
    double  *posterior_mean;      /* (n_chains, ndim, T)  Study.posteriorMeanValues (core.py:480-483)            */
    int64_t *abort_step;          /* (n_chains,)          -1, or the step i at which the normaliser was zero     */
    int32_t *abort_phase;         /* (n_chains,)          0 = forward (core.py:390-400), 1 = backward (442-452)  */
} blhip_result;

/* Timing of the last blhip_fit, measured with HIP events on the library's stream. */
typedef struct {
    double  forward_ms;           /* all forward-step launches                                                   */
    double  backward_ms;          /* all backward-step launches                                                  */
    double  accumulate_ms;        /* hyper-average accumulation launches                                         */
    double  total_ms;             /* whole call on the device timeline                                           */
    int64_t forward_launches;
    int64_t backward_launches;
    int64_t accumulate_launches;
    int64_t cells_per_launch;     /* grid cells x chains processed by one step launch (largest batch)            */
    int64_t batches;              /* number of chain batches the call was split into                             */
    int32_t fwd_kernel_variant;   /* which step kernel ran (library-internal id)                   */
    int32_t bwd_kernel_variant;
    /* what the launches of the call move and compute BY CONSTRUCTION (totals over all batches; the launch-per-step kernels
     * stream the state, the resident kernels keep it in LDS and touch HBM only for what the fit keeps + halo strips): the
     * figures bench.py turns into real HBM GB/s and fp64 TFLOP/s next to the streaming-equivalent rate of SURVEY 8(d) */
    double  fwd_hbm_bytes;        /* HBM bytes of all forward launches                                           */
    double  bwd_hbm_bytes;        /* HBM bytes of all backward launches (incl. a fold fused into them)           */
    double  fwd_flops;            /* fp64 flop of all forward launches, as executed (band products of the matrix-pipe
                                     kernels incl. their structural zeros, FMA = 2)                              */
    double  bwd_flops;
    int32_t resident_fallbacks;   /* batches of this call a resident launch gave up on (repeated launch-per-step) */
    int32_t resident_armed;       /* 1: the context will try the resident paths on its next eligible fit          */
    int32_t resident_fallback_reason;   /* (ABI v7) why the LAST such batch fell back: BLHIP_FALLBACK_*             */
    int32_t peer_copy_path;       /* (ABI v7) how blhip_accum_peer_reduce / _gather fetched the other contexts' slices since the
                                     last blhip_fit of THIS context: BLHIP_PEER_* (0: no peer merge)                   */
    int32_t resident_probe;       /* (ABI v8) the co-residency probe of this call: 0 = none ran (it runs on a context's first fit, when
                                     the resident paths are re-armed and at most once per second), 1 = every block of a one-block-per-CU
                                     grid with the resident kernels' footprint was on the chip at once, 2 = not (another process holds
                                     CUs, a partitioned GPU): the resident paths are parked WITHOUT paying a launch's time-out     */
    int32_t xcd_order;            /* (ABI v8) 1: the probe saw block b on XCD b % 8 (the both-axes kernels publish their exchange with
                                     plain stores that stay in that XCD's L2); 0: it did not -- write-through exchange instead    */
} blhip_timing;

/* blhip_timing.peer_copy_path */
#define BLHIP_PEER_NONE         0
#define BLHIP_PEER_SAME_DEVICE  1   /* both contexts on one GPU (test configuration): device-to-device copy                       */
#define BLHIP_PEER_DIRECT       2   /* hipMemcpyPeerAsync over xGMI with peer access                                                */
#define BLHIP_PEER_HOST_STAGED  3   /* no peer access / a failed peer copy / option peer_copy_mode = 1: through page-locked host memory */

/* blhip_timing.resident_fallback_reason */
#define BLHIP_FALLBACK_NONE       0
#define BLHIP_FALLBACK_GAVE_UP    1   /* a block waited longer than the bound for a peer block (blocks not all co-resident: a
                                         shared or partitioned GPU); the context parks the resident paths (resident_armed = 0) */
#define BLHIP_FALLBACK_RANGE      2   /* a lagged sum left (1e-150, 1e150) or was not finite (extreme outliers, a zero normaliser) */
#define BLHIP_FALLBACK_PREDICTION 3   /* the predicted posterior sums did not reproduce the reduced ones                          */
#define BLHIP_FALLBACK_FORCED     4   /* option resident_force_abort (tests)                                                      */
#define BLHIP_FALLBACK_BUSY       5   /* (ABI v8) no launch was tried: the co-residency probe found the chip shared (resident_probe = 2) */

/* ---- context ------------------------------------------------------------------------------------------------- */
int         blhip_abi_version(void);
int         blhip_device_count(void);
/* Diagnostics / test infrastructure (needs no GPU, no context): the kernel instantiations this library holds -- every __global__
 * function it can launch registers itself when the library is loaded -- and how often THIS PROCESS has launched each.  Text, one
 * line per instantiation, sorted by name: "<launches>\t<kernel name with its template arguments>\n".  Writes at most cap - 1 bytes
 * + NUL into buf (buf may be NULL); returns the length of the whole text (call again with a larger buffer if >= cap), < 0 on error.
 * The -m gpu suite asserts that no line starts with "0": every kernel that ships has been compared with the oracle. */
int64_t     blhip_kernel_census(char *buf, int64_t cap);
blhip_ctx  *blhip_create(int device);
void        blhip_destroy(blhip_ctx *ctx);
const char *blhip_last_error(blhip_ctx *ctx);        /* ctx may be NULL for errors of blhip_create              */
int         blhip_device_name(blhip_ctx *ctx, char *buf, int buflen);
int         blhip_set_option(blhip_ctx *ctx, const char *key, double value);   /* tuning knobs, see DESIGN.md    */
int         blhip_synchronize(blhip_ctx *ctx);

/* ---- the hot path --------------------------------------------------------------------------------------------- */
/* Runs n_chains independent forward(-backward) passes that share `problem` and differ in the hyper-parameter value
 * of each op: op_values is (n_chains, n_ops), row-major; entries of STATIC ops are ignored.
 * log_chain_weight (n_chains,) is only read with BLHIP_ACCUMULATE: log of the hyper-prior value of each chain
 * (core.py:1366); the library adds the chain's log-evidence itself.
 * Threads: a call with several batches of chains starts ONE helper thread that prepares the next batch's per-step programs
 * from `problem` / `op_values` (read-only, no HIP call) and is joined before the call returns -- also on errors; option
 * build_ahead = 0 keeps everything on the calling thread.  One context must not be used by two threads at once. */
int blhip_fit(blhip_ctx *ctx, const blhip_problem *problem, int64_t n_chains, const double *op_values,
              const double *log_chain_weight, uint32_t flags, blhip_result *result);

int blhip_last_timing(blhip_ctx *ctx, blhip_timing *out);

/* Calibration of the attainable HBM rate on THIS device ("calibrate the attainable peak on the box"): a 16-B-per-
 * lane streaming copy of `bytes` (read + write counted), `iterations` launches timed with HIP events -> GB/s. */
int blhip_bandwidth_probe(blhip_ctx *ctx, int64_t bytes, int iterations, double *gb_per_s);

/* ---- page-locked host memory for the big read-backs -------------------------------------------------------------------------------
 * posteriorSequence is a HOST array in the reference (core.py:356, 408): (T, G) doubles, 16 GiB for BASELINE C3.  Into pageable
 * memory the copy is staged by the runtime (23 GB/s measured); into page-locked memory it is ONE DMA at the PCIe rate.  The host
 * side allocates its result arrays here (bayesloop_amd/engine.py keeps one freed block for the next fit).  NULL on failure. */
void *blhip_host_alloc(size_t bytes);
void  blhip_host_free(void *p);

/* ---- posterior sequence of the last blhip_fit(..., BLHIP_KEEP_POSTERIOR) ---------------------------------------- */
/* Copies the normalised posteriors of steps [t0, t1) of one chain to host memory ((t1-t0) * G doubles). */
int blhip_posterior_read(blhip_ctx *ctx, int64_t chain, int64_t t0, int64_t t1, double *host_out);
/* Device address of chain 0 / step 0 and the strides in doubles (device-side consumers, e.g. RCCL). */
int blhip_posterior_devptr(blhip_ctx *ctx, void **devptr, int64_t *chain_stride, int64_t *step_stride);
int blhip_posterior_release(blhip_ctx *ctx);

/* ---- reductions of a device-resident posterior sequence (consumers of posteriorSequence: marginal parameter
 *      distributions, bayesloop/core.py:915, 979-980; time average, core.py:588, 886) -------------------------------
 * source: 0 = posterior of `chain` kept by the last blhip_fit(BLHIP_KEEP_POSTERIOR), 1 = the finalised accumulator.
 * keep_axis: the parameter whose marginal is wanted; host_out is (T, n[keep_axis]) probabilities (sums, no density). */
int blhip_posterior_marginal(blhip_ctx *ctx, int source, int64_t chain, int keep_axis, double *host_out);
/* host_out is (G,): mean over time steps of the posterior. */
int blhip_posterior_time_average(blhip_ctx *ctx, int source, int64_t chain, double *host_out);

/* ---- evidence-weighted average posterior (HyperStudy) ----------------------------------------------------------- */
/* The accumulator holds  A[t, cell] = sum_h exp(logE_h + log prior_h - log_ref) * max(post_h[t, cell], 1e-300)
 * in linear space with a running reference exponent `log_ref` (equal to the reference's log-space logaddexp
 * accumulation, core.py:1362-1366, up to rounding).  external_devptr, if not NULL, is caller-owned device memory of
 * T*G doubles (e.g. a torch tensor handed to RCCL); otherwise the library allocates it. */
int blhip_accum_begin(blhip_ctx *ctx, int64_t T, int64_t G, void *external_devptr);
int blhip_accum_state(blhip_ctx *ctx, double *log_ref, void **devptr, int64_t *n_folded);
/* Re-express the accumulator against a new reference exponent (>= current), e.g. the maximum over ranks, so that
 * accumulators of several GPUs can be summed (core.py:1339). */
int blhip_accum_rescale(blhip_ctx *ctx, double new_log_ref);
/* (ABI v8) Folds ONE chain whose posterior sequence the CALLER holds -- a hyper-grid point fitted through the transition-model plug-in
 * interface, where the state crosses the host at every step anyway (bayesloop_amd/core.py: Study._fitHostTransition) -- into the open
 * accumulator: posterior is (T, G) on the host, every row normalised; log_weight = logEvidence + log(hyper-prior value) of the chain
 * (core.py:1358-1366: acc = logaddexp(acc, log(max(post, 1e-300)) + log_weight); a non-finite log_weight contributes nothing).  The
 * sequence is uploaded and folded by the kernels that fold a stored batch. */
int blhip_accum_fold_host(blhip_ctx *ctx, const double *posterior, double log_weight);
/* Per-step normalisation (core.py:1379-1382) and posterior means (core.py:1416-1419); the normalised average stays
 * on the device and is read with blhip_accum_read.  posterior_mean: (ndim, T) or NULL. */
int blhip_accum_finalize(blhip_ctx *ctx, const blhip_problem *problem, double *posterior_mean);
int blhip_accum_read(blhip_ctx *ctx, int64_t t0, int64_t t1, double *host_out);
int blhip_accum_end(blhip_ctx *ctx);
/* Per-step sums of the (not yet finalised) accumulator, relative to its current reference exponent:
 * host_out (T, 1 + ndim) = [sum A, sum A grid_0, (sum A grid_1)] per step.  Sums of several ranks, each scaled by
 * exp(log_ref_rank - max log_ref), add up to the normalisers / posterior means of the merged average (core.py:1379-1382,
 * 1416-1419), so every rank can form them from ONE gather.  All zeros if nothing was folded. */
int blhip_accum_row_stats(blhip_ctx *ctx, const blhip_problem *problem, double *host_out);

/* ---- host-side algebra of the resident kernels, callable without a GPU (unit tests) ------------------------------------------------
 * The time-resident kernels divide step k not by the sum of step k - 1 but by something older (their sums cross the chip through
 * HBM); the host reconstructs the reference's normalisers (core.py:385) from the sums S_k they report:
 *   scheme 0 (blhip_resident.hpp):  s_k = 1 / S_(k-lag)                              norm_k = S_k / (S_(k-1) s_k)
 *   scheme 1 (blhip_chainres.hpp):  s_k = S_(k-lag-1) s_(k-lag) / S_(k-lag)          (s_k = 1 while k < lag, S_(-1) = 1)
 * sums: (T) in, the normalisers out (in place).  kinds (scheme 1, may be NULL): (T) source kinds, a step whose kind is not 0
 * (blk::SRC_PREV) restarted from a distribution of known mass: norm_k = S_k / s_k.  scales_out (may be NULL): (T) the scales s_k.
 * Returns 0, or 1 if a sum left (1e-150, 1e150) -- the fit then repeats the batch with the launch-per-step kernels. */
int blhip_host_unlag(int scheme, double *sums, int64_t T, int lag, const unsigned char *kinds, double *scales_out);

/* ---- multi-GPU exchange of a sharded hyper-study (HyperStudy.fit(nJobs > 1), core.py:1307-1340, 1443-1495) -------------
 * One process per GPU, one context per process; each rank fits its share of the hyper-grid points with blhip_fit and
 * no communication, then the ranks exchange results through RCCL (xGMI inside a node), which this library binds
 * directly: librccl.so.1 is dlopen'ed by the first blhip_comm_* call (override the path with BLHIP_RCCL_LIBRARY).
 *   rank 0:      blhip_comm_unique_id(id)  ->  the caller hands the 128 bytes to the other ranks (file, socket, ...)
 *   every rank:  blhip_comm_init(ctx, id, world, rank)                      (collective; ncclCommInitRank on ctx's device)
 *   after the fits, the ONE exchange:
 *                blhip_comm_allgather   packed rows [logEvidence | localEvidence (T) | abort step] per chain + a trailer
 *                                       [accumulator log_ref | row stats], `count` doubles per rank, host in / host out
 *                                       (staged through HBM; host_out is (world, count) in rank order)     core.py:1335-1337
 *                blhip_comm_reduce_accum  only when posteriors are wanted, after blhip_accum_rescale(max log_ref):
 *                                       ncclReduce(sum) of the (T, G) accumulators to `root`, in place       core.py:1338-1340
 *   blhip_comm_allreduce: small host vectors (barriers, max-over-ranks timing of bench.py). */
#define BLHIP_UNIQUE_ID_BYTES 128
enum { BLHIP_SUM = 0, BLHIP_MAX = 1, BLHIP_MIN = 2 };
int blhip_comm_unique_id(void *id_out /* BLHIP_UNIQUE_ID_BYTES */);      /* errors: blhip_last_error(NULL) */
int blhip_comm_init(blhip_ctx *ctx, const void *unique_id, int world, int rank);
int blhip_comm_info(blhip_ctx *ctx, int *world, int *rank, int *rccl_version);   /* world = 1 without a communicator */
int blhip_comm_allgather(blhip_ctx *ctx, const double *host_in, int64_t count, double *host_out);
int blhip_comm_allreduce(blhip_ctx *ctx, double *host_inout, int64_t count, int op);
int blhip_comm_reduce_accum(blhip_ctx *ctx, int root);
int blhip_comm_timing(blhip_ctx *ctx, double *reduce_ms);     /* HIP-event time of the last blhip_comm_reduce_accum (ms) */
int blhip_comm_destroy(blhip_ctx *ctx);

/* ---- the same merge inside ONE process that drives several GPUs (HyperStudy.fit(nJobs = N), core.py:1307-1340: one context per
 * device, one host thread per context; no RCCL): peer copies over xGMI.  Every accumulator involved must be open, of the same shape
 * and at the same reference exponent (blhip_accum_rescale), every context idle (blhip_synchronize); the caller orders the calls of
 * its threads with host barriers.  The merge is a reduce-scatter over time slices followed by a gather:
 *   blhip_accum_peer_reduce   dst.acc[row0:row1] += sum_i srcs[i].acc[row0:row1]   (rows fetched concurrently, one stream -- one
 *                             xGMI link -- per source; summed in list order: the same result on every run)
 *   blhip_accum_peer_gather   dst.acc[row0[i]:row1[i]] = srcs[i].acc[row0[i]:row1[i]]   for every i (concurrent copies)
 * at most NBS = 18 sources per call. */
int blhip_accum_peer_reduce(blhip_ctx *dst, blhip_ctx *const *srcs, int n_srcs, int64_t row0, int64_t row1);
int blhip_accum_peer_gather(blhip_ctx *dst, blhip_ctx *const *srcs, int n_srcs, const int64_t *row0, const int64_t *row1);

/* ---- carried states (OnlineStudy.step, core.py:2062-2226) ------------------------------------------------------------
 * A blhip_fit with BLHIP_CARRY leaves every chain's normalised filtered distribution of its last step in slot
 * `carry_slot` of the context; the next call with BLHIP_RESUME continues from it.  What the online study does with these
 * states after each step (core.py:2196-2212) are evidence-weighted sums over chains:
 *   mix = (accumulate ? mix : 0) + sum_j weights[j] * state_j      (transitionModelPosterior / marginalizedPosterior)
 * blhip_carry_read copies one chain's state (chain >= 0) or the mix buffer (chain = -1, slot ignored) to the host, (G,). */
int blhip_carry_mix(blhip_ctx *ctx, int slot, int64_t n_chains, const double *weights, int accumulate);
int blhip_carry_read(blhip_ctx *ctx, int slot, int64_t chain, double *host_out);
/* Restores carried states saved with blhip_carry_read (a pickled OnlineStudy that continues in another process / context,
 * reference fileIO.py:10-37): host_in is (n_chains, G), every row a normalised distribution. */
int blhip_carry_write(blhip_ctx *ctx, int slot, int64_t n_chains, int64_t G, const double *host_in);
int blhip_carry_release(blhip_ctx *ctx, int slot);   /* slot < 0: all slots and the mix buffer */

#ifdef __cplusplus
}
#endif
#endif /* BLHIP_H */
