"""
Every kernel instantiation libblhip.so holds, launched on the smallest problem that selects it and compared with the CPU oracle.

The library's kernels are templates -- ring length x tiles per wave x pass flavour x likelihood source, tile shape x pass x padding,
radius bucket x model x direction ... --, ~1 400 instantiations, selected at run time from the grid, the radii of the walks, the kind of
fit and the number of chains.  The parity suites (goldens, seeded configurations, bench workloads) exercise every template-argument
VALUE; this file walks the PRODUCT space: for each selector of the host code (bayesloop_amd/csrc/blhip.hip: launch_*, plan_*;
blhip_fit_paths.hpp; blhip_chain_tu.hip) it enumerates the tuples that selector can produce, builds the smallest study that produces
each, checks through the library's own registry (blhip_kernel_census, include/blhip.h) that exactly that kernel ran, and holds the
results to the parity bar (logE 1e-9, posteriors |dp| <= 1e-12 + 1e-9 p) against oracle/bl_oracle.py.

tests/test_zz_kernel_census.py then fails the -m gpu suite if the library holds an instantiation that no test of the session launched.
Reference semantics pinned: bayesloop/core.py:372-470 (Study.fit), :1349-1366 (hyper average), transitionModels.py:96-118 (random
walk), :632-662 (combined), :289-317 (change point).
"""
import numpy as np
import pytest

import bayesloop_amd as bl
import cases
import compare
import oracle_adapter as oa
from conftest import kernel_census
from test_gpu_parity import result_of, _ill_tol, _g2, _om2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def hip_engine():
    prev = bl.set_engine(None)
    eng = bl.get_engine()
    assert type(eng).__name__ == 'HipEngine'
    yield eng
    bl.set_engine(prev)


def _counts():
    return {name: c for c, name in kernel_census()}


def b(v):
    return 'true' if v else 'false'


class Options:
    """Engine options for the duration of a block (restored to the library's defaults afterwards)."""
    DEFAULT = dict(chain_resident=1, resident=1, chain_table=1, resident_table=1, mfma=1, mfma_h=1, mfma_h_max_cells=2.5e6, fast=1,
                   recurrence=1, chain1d=1, persist1d=1, fuse1d=8, chain_depad=1, mfma_S=0, fast_S=0, wide_h_fused_max=0)

    def __init__(self, **opts):
        self.opts = opts

    def __enter__(self):
        for k, v in self.opts.items():
            bl.get_engine().set_option(k, v)

    def __exit__(self, *exc):
        for k in self.opts:
            bl.get_engine().set_option(k, self.DEFAULT[k])


def run(c, expect, forbid_fallback=True, tol=None):
    """Fit the case, require that every kernel of `expect` was launched by it, compare everything the fit produced with the oracle."""
    before = _counts()
    S = cases.build(bl, c)
    kw = cases.fit_kwargs(c)
    S.fit(**kw)
    got = result_of(S, c)                # (before the census: materialising the posteriors launches the normalisation kernels)
    after = _counts()
    ran = sorted(k for k in after if after[k] > before.get(k, 0))
    missing = [k for k in expect if k not in ran]
    assert not missing, 'expected kernel(s) not launched: %s\nlaunched: %s\ntiming: %s' % (missing, ran, S.lastTiming)
    if forbid_fallback:
        assert S.lastTiming['resident_fallbacks'] == 0, S.lastTiming
    with np.errstate(all='ignore'):
        want = oa.run(c)
    gold = dict(logEvidence=want['logEvidence'], localEvidence=want['localEvidence'])
    for k in ('posteriorSequence', 'posteriorMeanValues', 'logEvidenceList', 'hyperParameterDistribution'):
        if k in want and want[k] is not None and k in got and len(np.atleast_1d(want[k])):
            gold[k] = np.asarray(want[k])
    case_tol = _ill_tol(S)
    if tol:
        case_tol = dict(case_tol or {}, **tol)
    compare.check(got, gold, compare.GPU_TOL, case_tol=case_tol)
    return S


# ---- the chain-resident kernels (blhip_chainres.hpp; selectors: plan_chainres, ChainRun::setup, blhip_chain_tu.hip: launch_k / launch_k_tab /
#      launch_fold2_k): ring length NK = (16 + 2 r0) / 4 for the band radius r0 = 4, 8, .. 80 of the widest chain of a launch (NK = 4: no stencil),
#      tiles per wave NTW = rows of the geometry / 128, forward / backward, storing or not (folding), padded or exact, tabulated likelihood --------

def _chain_geometry(ntw, pad):
    n0p = 128 * ntw
    return (n0p - 28, 20) if pad else (n0p, 16)


def _chain_sigmas(nk, n0, lo, hi):
    """Three walk widths whose SciPy radii int(4 sigma / lattice + 0.5) (transitionModels.py:108-111) are r0 - 1, r0 - 2, r0 - 3 for the
    band radius r0 of ring length nk; nk = 4: radius 0 (the walk is a copy, :113)."""
    lattice = (hi - lo) / (n0 - 1.0)
    if nk == 4:
        return [0.0, 1e-9 * lattice, 2e-9 * lattice]
    r0 = 2 * nk - 8
    return [(r0 - 1 - k) / 4.0 * lattice for k in range(3)]


CHAIN_SWEEP = [(False, ntw, pad, nk) for ntw in (1, 2, 3, 4, 8) for pad in (False, True) for nk in range(4, 46, 2)] + \
              [(True, ntw, pad, nk) for ntw, pad in ((1, False), (1, True), (2, False), (2, True), (3, False), (4, False)) for nk in range(4, 26, 2)]


@pytest.mark.parametrize('tab,ntw,pad,nk', CHAIN_SWEEP, ids=['%s-ntw%d-%s-nk%d' % ('table' if t else 'gauss', w, 'pad' if p else 'exact', k)
                                                              for t, w, p, k in CHAIN_SWEEP])
def test_chain_resident_kernel_instantiations(tab, ntw, pad, nk):
    n0, n1 = _chain_geometry(ntw, pad)
    seed = 7000 + 97 * nk + 11 * ntw + (5 if pad else 0) + (3 if tab else 0)
    if tab:
        lo, hi = -5.0, 5.0
        om = _om2('Laplace', ('mu', ('cint', lo, hi, n0)), ('b', ('oint', 0, 3, n1)))
        target = 'mu'
    else:
        lo, hi = -8.0, 8.0
        om = _g2(n0, n1)
        target = 'mean'
    sig = _chain_sigmas(nk, n0, lo, hi)
    T = 6
    k = 'blc::chain_kernel<%d, %d, %%s, %%s, %s, %s>' % (nk, ntw, b(pad), b(tab))
    hyper = dict(study='HyperStudy', data=('series', seed, T), om=om, tm=('GRW', 'sigma', sig, target, None))
    # evidence-only hyper-study: the forward kernel that stores nothing
    run(dict(hyper, fit=dict(evidenceOnly=True)), [k % ('false', 'false')])
    # full hyper-study: the storing forward kernel + the fold -- two chains per block on the geometries of <= 512 rows (three chains: one
    # pair and an odd chain), the one-chain folding kernel at 1024 rows and with a tabulated likelihood (padded + tabulated: stored and
    # folded by accumulate_pad_kernel)
    if tab:
        fold = k % ('true', 'true') if pad else k % ('true', 'false')
    elif ntw <= 4:
        fold = 'blc::chain_fold2_kernel<%d, %d, %s>' % (nk, ntw, b(pad))
    else:
        fold = k % ('true', 'false')
    run(hyper, [k % ('false', 'true'), fold])
    # a plain Study (one chain, posteriors handed out; padded grids: through the de-padding copy): the storing backward kernel
    # (1024 rows: on the exact geometry only -- padded full fits of one chain keep the launch-per-step kernels there)
    if not (ntw == 8 and pad):
        tm = ('Static',) if nk == 4 else ('GRW', 'sigma', sig[0], target, None)
        run(dict(study='Study', data=('series', seed + 1, T), om=om, tm=tm), [k % ('false', 'true'), k % ('true', 'true')])


# ---- walks on both parameters (blhip_chainax.hpp; selector: plan_chainres ax1, launch_chainax): ring lengths 8 .. 24 in steps of 4 (radius <= 8 .. 40
#      on either axis), square geometries of 128 / 256 / 512 (NTW 1 / 2 / 4; the 128 and 256 kernels are the padded ones for every grid) ----------

AX_SWEEP = [(ntw, pad, nk) for ntw, pad in ((1, True), (2, True), (4, False), (4, True)) for nk in (8, 12, 16, 20, 24)]


@pytest.mark.parametrize('ntw,pad,nk', AX_SWEEP, ids=['ntw%d-%s-nk%d' % (w, 'pad' if p else 'exact', k) for w, p, k in AX_SWEEP])
def test_both_axes_chain_resident_kernel_instantiations(ntw, pad, nk):
    n0, n1 = {(1, True): (100, 90), (2, True): (200, 180), (4, False): (512, 512), (4, True): (500, 400)}[(ntw, pad)]
    r0 = 2 * nk - 8
    lat0, lat1 = 16.0 / (n0 - 1.0), 4.0 / (n1 + 1.0)
    s1 = [(r0 - 1 - k) / 4.0 * lat0 for k in range(2)]
    s2 = (r0 - 3) / 4.0 * lat1
    T = 4
    seed = 8000 + 13 * nk + ntw + (7 if pad else 0)
    k = 'blc::chainax_kernel<%d, %d, %%s, %%s, %s>' % (nk, ntw, b(pad))
    hyper = dict(study='HyperStudy', data=('series', seed, T), om=_g2(n0, n1),
                 tm=('Combined', [('GRW', 's1', s1, 'mean', None), ('GRW', 's2', s2, 'std', None)]))
    run(dict(hyper, fit=dict(evidenceOnly=True)), [k % ('false', 'false')])
    run(hyper, [k % ('false', 'true'), k % ('true', 'false')])
    if nk > 8:
        # a plain Study: the storing backward kernel
        run(dict(study='Study', data=('series', seed + 1, T), om=_g2(n0, n1),
                 tm=('Combined', [('GRW', 's1', s1[0], 'mean', None), ('GRW', 's2', s2, 'std', None)])), [k % ('false', 'true'), k % ('true', 'true')])
    else:
        # (walks of radius <= 8 on both parameters: a single chain takes the time-resident kernel wherever its tiles fit the grid; what stores
        #  backward through these kernels is a batch that cannot fold inside them -- a change point inside filtering chains, transitionModels.py:300-312)
        run(dict(study='ChangepointStudy', data=('series_jump', seed + 1, T + 2, 3, 1.5), om=_g2(n0, n1),
                 tm=('Combined', [('GRW', 's1', s1[0], 'mean', None), ('GRW', 's2', s2, 'std', None), ('ChangePoint', 'tChange', [2, 4], None)])),
            [k % ('false', 'true'), k % ('true', 'true')])


# ---- the time-resident kernel (blhip_resident.hpp; selectors: plan_resident, launch_resident_t / _tab): tile shape x {evidence-only, full, forward-only}
#      x {whole tiles, padded last tile row / column} x {Gaussian recurrence, tabulated likelihood}; walks of radius <= 8 on both parameters ---------------

RES_GRIDS = {   # (tile rows, tile columns, padded) -> the smallest grid plan_resident gives that shape
    (64, 64, False): (64, 64), (64, 64, True): (88, 72),
    (32, 64, False): (32, 64), (32, 64, True): (48, 64),
    (32, 32, False): (32, 32), (32, 32, True): (48, 40),
    (128, 128, False): (1152, 1024), (128, 128, True): (1152, 960),
}
RES_SWEEP = [(tr, tc, pad, tab, kind) for (tr, tc, pad) in RES_GRIDS for tab in (False, True) for kind in ('evidence', 'full', 'forward')
             if not (tab and tr == 128) and not (tr == 128 and pad and kind == 'full')]


@pytest.mark.parametrize('tr,tc,pad,tab,kind', RES_SWEEP, ids=['%dx%d-%s-%s-%s' % (a, c, 'pad' if p else 'exact', 'table' if t else 'gauss', k)
                                                               for a, c, p, t, k in RES_SWEEP])
def test_time_resident_kernel_instantiations(tr, tc, pad, tab, kind):
    n0, n1 = RES_GRIDS[(tr, tc, pad)]
    seg = 32 if tr == 128 else 8
    T = 5 if tr < 128 else 3
    seed = 9000 + tr + 3 * tc + (5 if pad else 0) + (2 if tab else 0)
    if tab:
        om = _om2('Laplace', ('mu', ('cint', -5, 5, n0)), ('b', ('oint', 0, 3, n1)))
        names = ('mu', 'b')
        lat0, lat1 = 10.0 / (n0 - 1.0), 3.0 / (n1 + 1.0)
    else:
        om = _g2(n0, n1)
        names = ('mean', 'std')
        lat0, lat1 = 16.0 / (n0 - 1.0), 4.0 / (n1 + 1.0)
    tm = ('Combined', [('GRW', 's1', 7 / 4.0 * lat0, names[0], None), ('GRW', 's2', 5 / 4.0 * lat1, names[1], None)])      # radii 7 and 5
    fit = dict(evidence=dict(evidenceOnly=True), full={}, forward=dict(forwardOnly=True))[kind]
    k = 'blr::resident_kernel<%d, %d, %d, 8, %%s, %%d, %s, %s>' % (tr, tc, seg, b(pad), b(tab))
    if kind == 'evidence':
        expect = [k % ('false', 1)]
    elif kind == 'full':
        expect = [k % ('false', 0 if (pad or tab) else 2), k % ('true', 0)]
    else:
        expect = [k % ('false', 0 if (pad or tab or tr == 128) else 3)]
    run(dict(study='Study', data=('series', seed, T), om=om, tm=tm, fit=fit), expect)


# ---- the launch-per-step kernels of 2-D grids (blhip_mfma.hpp, blhip_fast.hpp; selectors: bucket_step, run_step, launch_mfma_*, launch_fast_*):
#      what a batch runs on when no resident path takes it (more than 1024 rows or columns, carried states, ...) -- here with the resident paths
#      switched off, so that the grids stay small.  Radius bucket of the walk on the first parameter (0, 8, .. 40) x a walk of radius <= 8 on
#      the second one or none x Gaussian likelihood by recurrence / by exp (a mean grid that is not equally spaced) / tabulated x direction;
#      the matrix-pipe kernels also in their LEAN form (whole tile groups inside the grid) ------------------------------------------------------------

def _uneven(lo, hi, n):
    """A parameter grid that is NOT equally spaced (core.py:161-166: its lattice constant is then 1)."""
    x = np.linspace(lo, hi, n)
    return list(x + 0.2 * (hi - lo) / (n - 1) * np.sin(np.arange(n)))


def _step_case(model, n0, n1, r0, h, seed, T=4):
    """Three chains in the radius bucket (r0 - 8, r0] on the first parameter; h: a walk of radius 5 on the second one."""
    if model == 'table':
        om = _om2('Laplace', ('mu', ('cint', -5, 5, n0)), ('b', ('oint', 0, 3, n1)))
        names, lat0, lat1 = ('mu', 'b'), 10.0 / (n0 - 1.0), 3.0 / (n1 + 1.0)
    elif model == 'exp':
        om = ('Gaussian', [('mean', _uneven(-8, 8, n0)), ('std', ('oint', 0, 4, n1))], 'default')
        names, lat0, lat1 = ('mean', 'std'), 1.0, 4.0 / (n1 + 1.0)
    else:
        om = _g2(n0, n1)
        names, lat0, lat1 = ('mean', 'std'), 16.0 / (n0 - 1.0), 4.0 / (n1 + 1.0)
    sig = [0.0, 1e-9 * lat0, 2e-9 * lat0] if r0 == 0 else [(r0 - 1 - k) / 4.0 * lat0 for k in range(3)]
    tms = [('GRW', 's1', sig, names[0], None)]
    if h:
        tms.append(('GRW', 's2', 5 / 4.0 * lat1, names[1], None))
    return dict(study='HyperStudy', data=('series', seed, T), om=om, tm=('Combined', tms))


STEP_MODELS = {'recurrence': (2, True), 'exp': (2, False), 'table': (100, False)}
STEP_SWEEP = [(m, r0, h) for m in STEP_MODELS for r0 in (0, 8, 16, 24, 32, 40) for h in (False, True)]


@pytest.mark.parametrize('model,r0,h', STEP_SWEEP, ids=['%s-r%d-%s' % (m, r, 'both' if h else 'one') for m, r, h in STEP_SWEEP])
def test_launch_per_step_2d_kernel_instantiations(model, r0, h):
    om, rec = STEP_MODELS[model]
    seed = 9500 + r0 + (1 if h else 0) + om
    nk = 4 if r0 == 0 else (16 + 2 * r0) // 4
    fast = 'blf::fast_step_kernel<%d, %%d, %d, %s, %s>' % (om, r0, b(h), b(rec))
    with Options(chain_resident=0, resident=0):
        # the vector-ALU streaming kernels: the radius-0 launches without a second walk by default; everything else forced (both-axes launches
        # beyond mfma_h_max_cells cells take them by themselves; one-axis launches with option mfma = 0)
        with Options(mfma=0):
            run(_step_case(model, 140, 90, r0, h, seed), [fast % 0, fast % 1])
        if r0 == 0 and not h:
            return
        # the matrix-pipe kernels: segments that do not divide the rows (140 rows) and, one-axis only, the LEAN form (128 rows, one segment)
        mf = 'blm::mfma_step_kernel<%d, %%d, %d, %s, %s, %%s>' % (om, nk, b(rec), b(h))
        run(_step_case(model, 140, 90, r0, h, seed + 50), [mf % (0, 'false'), mf % (1, 'false')])
        if not h:
            run(_step_case(model, 128, 64, r0, h, seed + 100), [mf % (0, 'true'), mf % (1, 'true')])


# ---- the generic LDS-tile kernel (blhip_kernels.hpp; selector: launch_step): observation model x {forward, forward with the means of a forward-only
#      fit, backward}; it runs what no other kernel takes -- here the dense zero-boundary kernel of an AlphaStableRandomWalk (transitionModels.py:158-260)
#      on 1-D grids and a RegimeSwitch clamp (:394-415) on a 2-D one ------------------------------------------------------------------------------------

GENERIC = {
    1: dict(om=('Poisson', [('rate', ('oint', 0, 6, 60))], 'default'), data=('coal', 9), target='rate'),
    2: dict(om=_g2(24, 20), data=('series', 9701, 5), target='mean'),
    3: dict(om=('GaussianMean', [('mean', ('cint', -4, 4, 50))], 'default'), data=('gm', 9702, 6), target='mean'),
    100: dict(om=('Bernoulli', [('p', ('oint', 0, 1, 40))], 'default'), data=np.array([1, 0, 1, 1, 0, 1], dtype=float), target='p'),
}


@pytest.mark.parametrize('om', sorted(GENERIC))
@pytest.mark.parametrize('kind', ['full', 'forward'])
def test_generic_step_kernel_instantiations(om, kind):
    g = GENERIC[om]
    if om == 2:
        tm, tol = ('Combined', [('GRW', 's', 0.3, g['target'], None), ('RS', 'log10pMin', -7, None)]), None
    else:
        tm, tol = ('AlphaStable', 'c', 0.2, 'alpha', 1.5, g['target']), dict(cases.FFT_TOL)
    c = dict(study='Study', data=g['data'], om=g['om'], tm=tm)
    k = 'blk::step_kernel<%d, %%d, %%s>' % om
    if kind == 'forward':
        run(dict(c, fit=dict(forwardOnly=True)), [k % (0, 'true')], tol=tol)
    else:
        run(c, [k % (0, 'false'), k % (1, 'true')], tol=tol)


# ---- clamps on 1-D grids inside the chain-resident kernel (blhip_chain1d.hpp CL = 2; selector: plan_geometry shift1d / clamp1d): RegimeSwitch in
#      front of and behind a walk (clamp modes 1 / 2), NotEqual (mode 3), one chain and a batch, every observation-model flavour of the kernel ----------

CLAMP1D = {
    1: dict(om=('Poisson', [('rate', ('oint', 0, 6, 70))], 'default'), data=('coal', 12), target='rate', s=0.25),
    3: dict(om=('GaussianMean', [('mean', ('cint', -4, 4, 90))], 'default'), data=('gm', 9721, 9), target='mean', s=0.3),
    100: dict(om=('Bernoulli', [('p', ('oint', 0, 1, 50))], 'default'), data=np.array([1, 0, 1, 1, 0, 1, 1, 1, 0], dtype=float), target='p', s=0.05),
}
CLAMP_MODELS = {
    'walk_then_switch': lambda g: ('Combined', [('GRW', 's', g['s'], g['target'], None), ('RS', 'log10pMin', -5, None)]),      # clamp after the stencil
    'switch_then_walk': lambda g: ('Combined', [('RS', 'log10pMin', -5, None), ('GRW', 's', g['s'], g['target'], None)]),      # clamp on the source
    'switch_alone': lambda g: ('RS', 'log10pMin', -4, None),
    'not_equal': lambda g: ('NE', 'log10pMin', -6, None),
}


@pytest.mark.parametrize('model', sorted(CLAMP_MODELS))
@pytest.mark.parametrize('om', sorted(CLAMP1D))
def test_clamps_inside_the_one_dimensional_chain_resident_kernel(om, model):
    g = CLAMP1D[om]
    k = 'bl1c::chain1d_kernel<%d, %%s, 1, 2>' % om
    c = dict(study='Study', data=g['data'], om=g['om'], tm=CLAMP_MODELS[model](g))
    S = run(c, [k % 'false', k % 'true'])
    assert S.lastTiming['fwd_kernel_variant'] == 9 and S.lastTiming['bwd_kernel_variant'] == 9, S.lastTiming
    run(dict(c, fit=dict(forwardOnly=True)), [k % 'false'])


CLAMP1D_LONG = {      # rows longer than a block: two adjacent cells per thread (M = 2), even and odd lengths
    1: dict(om=('Poisson', [('rate', ('oint', 0, 6, 701))], 'default'), data=('coal', 10), target='rate', s=0.12),
    3: dict(om=('GaussianMean', [('mean', ('cint', -4, 4, 600))], 'default'), data=('gm', 9731, 8), target='mean', s=0.2),
    100: dict(om=('Bernoulli', [('p', ('oint', 0, 1, 1023))], 'default'), data=np.array([1, 0, 1, 1, 0, 1, 1], dtype=float), target='p', s=0.02),
}


@pytest.mark.parametrize('model', ['walk_then_switch', 'switch_then_walk', 'not_equal'])
@pytest.mark.parametrize('om', sorted(CLAMP1D_LONG))
def test_clamps_on_long_rows_two_cells_per_thread(om, model):
    g = CLAMP1D_LONG[om]
    k = 'bl1c::chain1d_kernel<%d, %%s, 2, 2>' % om
    run(dict(study='Study', data=g['data'], om=g['om'], tm=CLAMP_MODELS[model](g)), [k % 'false', k % 'true'])


def test_clamped_batches_on_the_one_dimensional_chain_resident_kernel():
    """Hyper-studies over the clamp level and the walk's width: a batch of chains (five and more share one likelihood table: the kernel's
    tabulated flavour), evidence-only too."""
    g = CLAMP1D[1]
    k = 'bl1c::chain1d_kernel<100, %s, 1, 2>'
    c = dict(study='HyperStudy', data=('coal', 20), om=g['om'],
             tm=('Combined', [('GRW', 's', ('cint', 0.1, 0.5, 3), 'rate', None), ('RS', 'log10pMin', [-7, -4], None)]))
    S = run(c, [k % 'false', k % 'true'])
    assert S.lastTiming['fwd_kernel_variant'] == 9 and S.lastTiming['bwd_kernel_variant'] == 9, S.lastTiming
    run(dict(c, fit=dict(evidenceOnly=True)), [k % 'false'])
    c = dict(study='HyperStudy', data=('coal', 16), om=g['om'], tm=('NE', 'log10pMin', ('cint', -7, -3, 5), None))
    run(c, [k % 'false', k % 'true'])


# ---- 1-D batches (blhip_chain1d.hpp; selector: plan_geometry chain1d, launch_chain1d): the shared likelihood table of a GaussianMean batch, and
#      Deterministic steps (transitionModels.py:548-606) in a batch too small for it ----------------------------------------------------------------------

def test_one_dimensional_gaussian_mean_batches():
    om = ('GaussianMean', [('mean', ('cint', -4, 4, 300))], 'default')
    # five widths: the batch shares ONE (T, n) likelihood table, built by lik1d_table_kernel<GaussianMean>
    run(dict(study='HyperStudy', data=('gm', 9711, 12), om=om, tm=('GRW', 'sigma', ('cint', 0.05, 0.4, 5), 'mean', None)),
        ['bl1c::lik1d_table_kernel<3>', 'bl1c::chain1d_kernel<100, false, 1, 0>', 'bl1c::chain1d_kernel<100, true, 1, 0>'])
    # two chains with a drift (a break point at two candidate positions): spline shifts, the likelihood evaluated in the kernel
    run(dict(study='ChangepointStudy', data=('gm', 9712, 10), om=om,
             tm=('Serial', [('Static',), ('BreakPoint', 't_break', [3, 6], None), ('Deterministic', 'drift', 'mean')])),
        ['bl1c::chain1d_kernel<3, false, 1, 1>', 'bl1c::chain1d_kernel<3, true, 1, 1>'], tol=dict(cases.FFT_TOL))


def test_bandwidth_probe_kernels():
    """blhip_bandwidth_probe (the calibrated peak beside the 8 TB/s spec in bench.py's roofline): copy and fill streams, both variants."""
    before = _counts()
    gbs = bl.get_engine().bandwidth_probe(1 << 28, 3)
    after = _counts()
    for k in ('blk::copy16_kernel<false>', 'blk::copy16_kernel<true>', 'blk::fill16_kernel<false>', 'blk::fill16_kernel<true>'):
        assert after[k] > before[k], k
    assert 1000.0 < gbs < 9000.0, gbs
