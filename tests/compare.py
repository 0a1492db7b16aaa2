"""Shared result comparison: a result dict (oracle or product) against a golden fixture."""
from __future__ import annotations

import numpy as np

import cases

# The parity bar of BASELINE.json / SURVEY.md section 8(d), float64:
#   log-evidence: 1e-9 relative;  posteriors: |dp| <= 1e-12 + 1e-9 * p
GPU_TOL = dict(logE_rtol=1e-9, post_rtol=1e-9, post_atol=1e-12, small_rtol=1e-9, small_atol=1e-12)
# the oracle restates the reference operation by operation -> rounding-level agreement
ORACLE_TOL = dict(logE_rtol=1e-13, post_rtol=1e-11, post_atol=1e-300, small_rtol=1e-11, small_atol=1e-300)


BAR_SMALL_RTOL = GPU_TOL['small_rtol']
# localEvidence entries compared in this session: at the bar / at a registered looser tolerance / NaN on both sides (0/0, core.py:463)
# local_partial: loosened entries whose sum over the cells with a NORMAL likelihood value was additionally compared at the bar
COUNTS = dict(local_at_bar=0, local_loosened=0, local_nan=0, local_partial=0)


def _close(a, b, rtol, atol, what):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), '%s: NaN pattern differs (%d vs %d NaNs)' % (what, nan_a.sum(), nan_b.sum())
    inf_a, inf_b = np.isinf(a), np.isinf(b)
    assert np.array_equal(inf_a, inf_b) and np.array_equal(a[inf_a], b[inf_b]), '%s: inf pattern differs' % what
    ok = ~(nan_a | inf_a)
    err = np.abs(a[ok] - b[ok]) - (atol + rtol * np.abs(b[ok]))
    if err.size and err.max() > 0:
        k = np.argmax(err)
        raise AssertionError('%s: max excess %.3e (got %r, want %r)' % (what, err.max(), a[ok][k], b[ok][k]))


def check(res, gold, tol, aborted_ok=True, case_tol=None):
    """res: dict with the attribute names of the reference's study objects."""
    gl = float(gold['logEvidence'])
    rl = float(res['logEvidence'])
    if not np.isfinite(gl):
        assert rl == gl, 'logEvidence %r vs %r' % (rl, gl)
        if 'logEvidenceList' not in gold:
            return                       # aborted Study.fit: remaining attributes are uninitialised in the reference
    else:
        assert abs(rl - gl) <= tol['logE_rtol'] * abs(gl), 'logEvidence %r vs %r (rel %.2e)' % (
            rl, gl, abs(rl - gl) / abs(gl))
    tol_is_gpu = tol is GPU_TOL
    local_rtol = max(tol['small_rtol'], (case_tol or {}).get('local_rtol', 0.0)) if tol_is_gpu else tol['small_rtol']
    # a case may state the absolute round-off floor of the REFERENCE itself (e.g. its FFT convolution) for posteriors
    tol = dict(tol, post_atol=max(tol['post_atol'], (case_tol or {}).get('post_atol', 0.0)),
               logE_rtol=max(tol['logE_rtol'], (case_tol or {}).get('logE_rtol', 0.0)),
               post_rtol=max(tol['post_rtol'], (case_tol or {}).get('post_rtol', 0.0)),
               small_rtol=max(tol['small_rtol'], (case_tol or {}).get('small_rtol', 0.0)))
    # a loosened local_rtol applies ONLY to the steps the case marks (local_loose_steps: denormal likelihood cells, the registered
    # exception ILL_LOCAL_EVIDENCE); all other entries keep the bar.  COUNTS: how many entries were compared at which tolerance
    loose = (case_tol or {}).get('local_loose_steps') if tol_is_gpu else None
    got_l, want_l = np.asarray(res['localEvidence'], dtype=float), np.asarray(gold['localEvidence'], dtype=float)
    if loose is not None and local_rtol > tol['small_rtol'] and got_l.shape == want_l.shape and got_l.shape[-1:] == np.shape(loose):
        loose = np.asarray(loose, dtype=bool)
        _close(got_l[..., ~loose], want_l[..., ~loose], tol['small_rtol'], tol['small_atol'], 'localEvidence')
        _close(got_l[..., loose], want_l[..., loose], local_rtol, tol['small_atol'], 'localEvidence (steps with denormal likelihood cells)')
        n_loose = int(np.isfinite(want_l[..., loose]).sum())
        COUNTS['local_loosened'] += n_loose
        COUNTS['local_at_bar'] += int(np.isfinite(want_l).sum()) - n_loose
        if (case_tol or {}).get('local_pinned'):        # hyper-studies: every chain's sum over the well-conditioned cells was compared at the bar (the caller did it)
            COUNTS['local_partial'] += n_loose
        # What the loosened entries leave open, pinned at the bar: localEvidence = 1 / (sum(post / L) dV) (core.py:463) is ill-conditioned
        # only through the cells whose likelihood is DENORMAL (1 .. 52 significant bits); the sum over all other cells, formed the
        # same way on both sides from the step's posterior and likelihood, must agree to 1e-9.
        liks = (case_tol or {}).get('local_lik')
        if liks and 'posteriorSequence' in gold and 'posteriorSequence' in res:
            pg, pw = np.asarray(res['posteriorSequence'], dtype=float), np.asarray(gold['posteriorSequence'], dtype=float)
            for t, L in liks.items():
                if t >= len(pw) or pw[t].shape != np.shape(L) or not np.isfinite(want_l[..., t]).all():
                    continue
                normal = L >= 2.2250738585072014e-308
                with np.errstate(all='ignore'):
                    sg, sw = float(np.sum(pg[t][normal] / L[normal])), float(np.sum(pw[t][normal] / L[normal]))
                assert abs(sg - sw) <= tol['small_rtol'] * abs(sw) + 1e-300, \
                    'localEvidence step %d: sum(post / L) over the cells with a normal likelihood value %r vs %r' % (t, sg, sw)
                COUNTS['local_partial'] += 1
    else:
        _close(got_l, want_l, local_rtol, tol['small_atol'], 'localEvidence')
        COUNTS['local_loosened' if local_rtol > BAR_SMALL_RTOL else 'local_at_bar'] += int(np.isfinite(want_l).sum())
    COUNTS['local_nan'] += int(np.isnan(want_l).sum())
    if 'posteriorMeanValues' in gold:
        _close(res['posteriorMeanValues'], gold['posteriorMeanValues'], tol['small_rtol'], 1e-11, 'posteriorMeanValues')
    if 'posteriorSequence' in gold:
        _close(res['posteriorSequence'], gold['posteriorSequence'], tol['post_rtol'], tol['post_atol'],
               'posteriorSequence')
    if 'posteriorRows' in gold:
        post = np.asarray(res['posteriorSequence'])
        rows = gold['posteriorRowsIndex']
        stride = gold['posteriorRowsStride'] if 'posteriorRowsStride' in gold else [1] * (post.ndim - 1)
        sub = post[rows][(slice(None),) + tuple(slice(None, None, int(s)) for s in stride)]
        _close(sub, gold['posteriorRows'], tol['post_rtol'], tol['post_atol'], 'posteriorRows')
        if 'marginalSequence0' in gold:
            _close(post.sum(axis=2), gold['marginalSequence0'], tol['post_rtol'], tol['post_atol'], 'marginal0')
            _close(post.sum(axis=1), gold['marginalSequence1'], tol['post_rtol'], tol['post_atol'], 'marginal1')
    for key in ('logEvidenceList',):
        if key in gold:
            g = gold[key]
            r = np.asarray(res[key], dtype=float)
            fin = np.isfinite(g)
            assert np.array_equal(fin, np.isfinite(r)), key + ': finiteness differs'
            assert np.all(np.abs(r[fin] - g[fin]) <= tol['logE_rtol'] * np.abs(g[fin])), key
    for key in ('hyperParameterDistribution', 'flatHyperPriorValues', 'hyperGridValues', 'hyperGridConstant'):
        if key in gold and key in res:
            _close(res[key], gold[key], max(tol['small_rtol'], 1e-9 if key == 'hyperParameterDistribution' else 0),
                   1e-15, key)
    if 'mask' in gold and 'mask' in res:
        assert np.array_equal(res['mask'], gold['mask'])
