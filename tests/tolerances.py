"""
The parity bar and EVERY exception to it, in one place (DESIGN.md section 6 lists the same entries; tests/test_tolerance_registry.py
fails when a test grants itself a tolerance that is not registered here, or when DESIGN.md does not name a registered exception).

Bar (BASELINE.json / SURVEY.md 8d, float64):  log-evidence 1e-9 relative;  posteriors |dp| <= 1e-12 + 1e-9 p  (compare.GPU_TOL).
"""
BAR = dict(logE_rtol=1e-9, post_rtol=1e-9, post_atol=1e-12, small_rtol=1e-9, small_atol=1e-12)

# one backward local-evidence entry of a step whose grid holds DENORMAL likelihood values (see ILL_LOCAL_EVIDENCE below)
ILL_LOCAL_RTOL = 5e-3
# the reference's own FFT / recursive-prefilter round-off (~1e-17 ABSOLUTE) -- a floor for the ORACLE-vs-reference comparison (whose
# absolute tolerance is otherwise 1e-300); on the GPU it is inside the bar (1e-15 < 1e-12)
FFT_TOL = dict(post_atol=1e-15, post_rtol=1e-9, logE_rtol=1e-12)
# seeded Deterministic-model fuzz (tests/random_cases.py)
DETERMINISTIC_FUZZ_TOL = dict(FFT_TOL, post_rtol=2e-8, logE_rtol=1e-10, small_rtol=2e-8)
# the fixture cases.py: wide_filter_2d
WIDE_FILTER_2D_TOL = dict(local_rtol=1e-3)

# the reference's published break-point study (bench.py: coal_breakpoints): see COAL_NOISE_CHAINS below
COAL_NOISE_TOL = dict(noise_chains=2, noise_weight_max=1e-6, logE_rtol=5e-9, post_atol=3e-4, mean_atol=1e-3, duration_rtol=5e-6)
# bench.py's parity gate: the bound on the user-visible S.logEvidence per workload where it is not the 1e-9 bar (every entry is a
# registered exception below)
BENCH_LOG_EVIDENCE_BOUND = {'coal_breakpoints': COAL_NOISE_TOL['logE_rtol']}

EXCEPTIONS = {
    'COAL_NOISE_CHAINS': dict(
        value=COAL_NOISE_TOL, above_bar=True,
        where='tests/test_gpu_parity.py: test_the_references_published_break_point_study_at_full_size (and bench.py extra.coal_breakpoints): '
              'at most 2 of the 23 400 chains may differ in WHETHER they stop with a non-positive normaliser (core.py:442-452) -- rounds 1 - 4: '
              '14; every chain that is finite on both sides keeps the 1e-9 bar (observed 6e-16), and so do the evidence of the average model and the '
              'hyper-parameter / duration distributions formed with the reference\'s stop pattern; the study\'s own logEvidence, which the '
              'differing chains enter, within 5e-9 (observed 6.0e-10 -- inside the bar; rounds 1 - 4: 8e-6); those chains\' WEIGHT in the average model '
              '(computed in the test) observed 4e-8, bounded at 1e-6 (rounds 1 - 4: 1e-3); the average posterior 3e-4 absolute (observed 5.8e-5: '
              'the one chain of weight 4e-8 that the reference keeps is renormalised noise with cell values up to ~1400 of either sign), its means '
              '1e-3 (observed 1.7e-4, grid 0 .. 6), the duration distribution 5e-6 relative (observed 6.9e-7)',
        seeds='the 14 chains that stop in the reference (slope -2.0 with break-points 3 .. 5 years apart, slope -1.52 with 1874 / 1878 and '
              '1879 / 1883) stop here too; ONE more chain of the slope -1.52 family stops here',
        reason='transitionModels.py:586-606: these chains shift their backward message by 250 - 334 grid cells per step, i.e. OFF the grid; '
               'what scipy.ndimage.shift leaves is the tail of its recursive spline prefilter, which the reference RENORMALISES to sum 1 '
               '(:603): its sign decides whether sum(alpha * beta) > 0 holds.  Round 5 restated the recursion bit for bit in the oracle '
               '(oracle/spline_iir.c: the oracle reproduces the reference\'s pattern chain for chain and its logEvidence to the last bit) and '
               'runs it on the device (blk::spline_prefilter_wave): all 14 stop.  The remaining difference is rounding noise in the '
               'reference itself: tests/test_oracle_golden.py shows that +-2 ulp on the prefilter\'s input leaves the 14 stopped and stops 0 - 2 '
               'further chains of the slope -1.52 family, moving the reference\'s own logEvidence by up to 8.9e-10 relative'),
    'ILL_LOCAL_EVIDENCE': dict(
        value=dict(local_rtol=ILL_LOCAL_RTOL), above_bar=True,
        where='tests/test_gpu_parity.py: seeded random configurations / resident-kernel cases, ONLY the localEvidence entries of steps '
              'whose likelihood has denormal cells (0 < L < 2.2e-308): a per-step mask built by _ill_tol(); steps with exact zeros '
              'only are NaN on both sides; the session summary prints how many entries were compared at which tolerance',
        seeds='3 of 6000 configurations of random_case (e.g. seed 2641): 1.6e-3 relative in ONE backward localEvidence entry -- the '
              'tolerance is 3 x that (round 3: 2e-2); for single-chain studies the sum over the cells with a NORMAL likelihood value is '
              'additionally compared at 1e-9 on both sides (compare.check: local_lik), which pins everything but the denormal cells',
        reason='core.py:463 localEvidence = 1 / sum(post / L): a denormal L carries 1..52 significant bits, post / L at such a cell '
               'can dominate the sum, so the reference value itself is defined to a few digits only; every other number of those '
               'cases keeps the 1e-9 bar'),
    'WIDE_FILTER_2D': dict(
        value=WIDE_FILTER_2D_TOL, above_bar=True,
        where='tests/cases.py: wide_filter_2d (golden fixture), backward localEvidence only',
        seeds='fixture wide_filter_2d (std values down to 0.08 put denormal likelihood values on the grid at step 0)',
        reason='same conditioning as ILL_LOCAL_EVIDENCE: any change of operation order moves the golden value at the 1e-4 level'),
    'DETERMINISTIC_FUZZ': dict(
        value=DETERMINISTIC_FUZZ_TOL, above_bar=True,
        where='tests/random_cases.py: random_model_case(kind == "deterministic") -- posteriors / means / local evidence 2e-8; the '
              'log-evidence stays at 1e-10',
        seeds='1 of 12 000 configurations (random_model_case seed 5 + 13 k family) at 8e-10, none above 2e-8',
        reason='transitionModels.py:581-602: a dozen cubic-spline shifts of a distribution that runs into the grid edge leave ringing; '
               'the reference renormalises by a sum that is 1e-2 .. 1e-3 of the mass (or negative), which amplifies any rounding '
               'difference by that factor per step'),
    'FFT_FLOOR': dict(
        value=FFT_TOL, above_bar=False,
        where='tests/cases.py (AlphaStable / Deterministic fixtures), tests/random_cases.py (alphastable): absolute floor 1e-15 of the '
              'oracle-vs-reference and HIP-vs-golden posterior comparison',
        seeds='kat_alphastable, alphastable_2d_hyper, alphastable_std, deterministic_* fixtures',
        reason='scipy.signal.fftconvolve / scipy.ndimage.shift round-off of the reference is ~1e-17 absolute, not relative to the '
               '(possibly 1e-200) posterior value'),
}
