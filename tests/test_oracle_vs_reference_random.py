"""Pins the oracle on SEEDED RANDOM configurations by running the real reference next to it.

Only where the reference is mounted (the build container: /root/reference); skipped elsewhere, e.g. on the GPU box.  The
reference is imported, never copied; the two harness shims of tests/golden/gen_golden.py apply (``numpy.math``, numpy-2-safe
attribute reads).  The same generators drive the GPU parity tests (tests/test_gpu_parity.py: HIP path vs oracle), so
random configuration -> reference == oracle (here) and oracle == HIP path (there).
"""
import contextlib
import io
import math
import os
import sys
import warnings

import numpy as np
import pytest

import cases
import oracle_adapter as oa
import random_cases

REFERENCE = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'bayesloop')), reason='reference not mounted')


@pytest.fixture(scope='module')
def ref():
    sys.path.insert(0, REFERENCE)
    np.math = math                      # observationModels.py:502 calls np.math.factorial (gone in numpy 2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import bayesloop
    yield bayesloop
    sys.path.remove(REFERENCE)


def _close(a, b, rtol, atol, what):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, what
    assert np.array_equal(np.isnan(a), np.isnan(b)), what + ': NaN pattern'
    ok = np.isfinite(a) & np.isfinite(b)
    assert np.array_equal(a[~ok & ~np.isnan(a)], b[~ok & ~np.isnan(b)]), what + ': inf pattern'
    assert np.all(np.abs(a[ok] - b[ok]) <= atol + rtol * np.abs(b[ok])), '%s: max diff %.3e' % (what, np.abs(a[ok] - b[ok]).max())


def _check(ref, c, tol=None):
    data = np.asarray(cases.make_data(c['data']))
    if c['om'][0] == 'Poisson' and data.dtype.kind == 'f':
        pytest.skip('the reference needs integer counts (factorial)')
    S = cases.build(ref, c)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        S.fit(**cases.fit_kwargs(c))
        w = oa.run(c)
    a, b = float(S.logEvidence), float(w['logEvidence'])
    assert a == b or abs(a - b) <= 1e-9 * abs(a), 'logEvidence %r (reference) vs %r (oracle)' % (a, b)
    stopped = not np.isfinite(a) or ('logEvidenceList' in w and not np.all(np.isfinite(np.asarray(w['logEvidenceList'], dtype=float))))
    if 'logEvidenceList' in w and len(np.atleast_1d(S.logEvidenceList)) > 1:
        _close(S.logEvidenceList, w['logEvidenceList'], 1e-9, 0.0, 'logEvidenceList')
        _close(S.hyperParameterDistribution, w['hyperParameterDistribution'], 1e-8, 1e-300, 'hyperParameterDistribution')
    if stopped:
        return                           # the reference leaves np.empty() garbage behind a zero-normaliser stop (core.py:360, :399)
    _close(S.localEvidence, w['localEvidence'], 1e-7, 0.0, 'localEvidence')
    if not cases.fit_kwargs(c).get('evidenceOnly') and w.get('posteriorSequence') is not None:
        atol = (tol or {}).get('post_atol', 0.0)
        _close(S.posteriorSequence, w['posteriorSequence'], 1e-9, max(1e-300, atol), 'posteriorSequence')
        _close(S.posteriorMeanValues, w['posteriorMeanValues'], 1e-9, 1e-12, 'posteriorMeanValues')


@pytest.mark.parametrize('seed', range(40))
def test_oracle_equals_reference_on_random_grw_studies(ref, seed):
    _check(ref, random_cases.random_case(seed))


@pytest.mark.parametrize('seed', range(52))
def test_oracle_equals_reference_on_random_model_zoo(ref, seed):
    c, tol = random_cases.random_model_case(seed)
    _check(ref, c, tol)


@pytest.mark.parametrize('seed', range(24))
def test_oracle_equals_reference_on_random_wide_walks(ref, seed):
    """Walks on the second parameter with radii 9 .. 64 and on the first one with radii 41 .. 128 (the pre-pass kernels on the GPU side)."""
    _check(ref, random_cases.random_wide_axis1_case(seed))


@pytest.mark.parametrize('seed', range(40))
def test_oracle_equals_reference_on_random_hyper_studies(ref, seed):
    _check(ref, random_cases.random_hyper_case(seed))


@pytest.mark.parametrize('seed', range(40))
def test_oracle_equals_reference_on_random_online_studies(ref, seed):
    c = random_cases.random_online_case(seed)
    S = cases.build_online(ref, c)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for d in cases.online_data(c):
            S.step(d)
        w = oa.run_online(c)
    a, b = float(S.logEvidence), float(w['logEvidence'])
    assert a == b or abs(a - b) <= 1e-9 * abs(a), 'logEvidence %r (reference) vs %r (oracle)' % (a, b)
    for key in ('posteriorSequence', 'posteriorMeanValues', 'transitionModelSequence', 'localTransitionModelSequence'):
        _close(np.asarray(getattr(S, key), dtype=float), np.asarray(w[key], dtype=float), 1e-9, 1e-13, key)
    for i in range(len(S.transitionModels)):
        _close(np.asarray([h[i] for h in S.hyperParameterSequence], dtype=float),
               np.asarray([h[i] for h in w['hyperParameterSequence']], dtype=float), 1e-9, 1e-13, 'hyperParameterSequence')
