"""Pins the CPU oracle: every golden vector generated from the reference, and every known-answer value the
reference's own tests hold for the hot path."""
import numpy as np
import pytest

import cases
import compare
import oracle_adapter as oa

FAST = [k for k, c in cases.CASES.items() if not c.get('slow')]
SLOW = [k for k, c in cases.CASES.items() if c.get('slow')]


@pytest.mark.parametrize('case', FAST)
def test_oracle_matches_reference_golden(case):
    with np.errstate(all='ignore'):
        res = oa.run(case)
    compare.check(res, oa.load_golden(case), compare.ORACLE_TOL, case_tol=cases.CASES[case].get('tol'))


@pytest.mark.slow
@pytest.mark.parametrize('case', SLOW)
def test_oracle_matches_reference_golden_slow(case):
    with np.errstate(all='ignore'):
        res = oa.run(case)
    compare.check(res, oa.load_golden(case), compare.ORACLE_TOL, case_tol=cases.CASES[case].get('tol'))


@pytest.mark.parametrize('case', [k for k in FAST if 'kat' in cases.CASES[k]])
def test_oracle_matches_reference_test_values(case):
    """The hard-coded expectations of the reference's own tests (same tolerances as there, or tighter)."""
    c = cases.CASES[case]
    with np.errstate(all='ignore'):
        res = oa.run(case)
    np.testing.assert_almost_equal(res['logEvidence'], c['kat'], decimal=c.get('kat_decimal', 5))
    if 'kat_hpd' in c:
        np.testing.assert_allclose(res['hyperParameterDistribution'] * np.prod(res['hyperGridConstant']),
                                   c['kat_hpd'], rtol=1e-5)


def test_golden_holds_reference_prior_and_grid():
    for case in FAST:
        g = oa.load_golden(case)
        res = oa.run(case) if cases.CASES[case]['study'] == 'Study' else None
        if res is None:
            continue
        np.testing.assert_array_equal(res['grid'].marginal[0], g['marginal0'])
        np.testing.assert_allclose(res['prior'], g['prior'], rtol=1e-15, atol=0)
