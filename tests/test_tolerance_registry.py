"""Every tolerance a test grants itself beyond the parity bar must be a registered exception (tests/tolerances.py) and be
named in DESIGN.md section 6."""
import os
import re

import numpy as np

import cases
import compare
import random_cases
import tolerances

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REGISTERED = [e['value'] for e in tolerances.EXCEPTIONS.values()]


def test_bar_is_the_stated_one():
    assert compare.GPU_TOL == tolerances.BAR
    assert tolerances.BAR['logE_rtol'] == 1e-9 and tolerances.BAR['post_rtol'] == 1e-9 and tolerances.BAR['post_atol'] == 1e-12


def test_fixture_cases_only_use_registered_exceptions():
    for name, c in list(cases.CASES.items()) + list(getattr(cases, 'ONLINE_CASES', {}).items()):
        if c.get('tol') is not None:
            assert c['tol'] in REGISTERED, 'case %s grants itself an unregistered tolerance %r' % (name, c['tol'])


def test_seeded_generators_only_use_registered_exceptions():
    for gen in (random_cases.random_model_case,):
        for seed in range(400):
            out = gen(seed)
            tol = out[1] if isinstance(out, tuple) else None
            if tol is not None:
                assert tol in REGISTERED, '%s(%d) grants itself an unregistered tolerance %r' % (gen.__name__, seed, tol)


def test_test_sources_define_no_private_tolerances():
    """Numeric tolerances looser than the bar in the parity tests must come from tests/tolerances.py: no `case_tol=dict(...)` /
    `tol = dict(...)` literals with numbers in them outside the registry."""
    pat = re.compile(r'(case_tol|tol)\s*=\s*dict\(([^)]*)\)')
    allowed = {'local_rtol=ILL_LOCAL_RTOL', 'local_rtol=ILL_LOCAL_RTOL, local_loose_steps=loose'}
    for fn in ('test_gpu_parity.py', 'test_online.py', 'test_optimize.py', 'test_reference_expectations.py', 'random_cases.py', 'cases.py', 'test_kernel_sweep.py'):
        src = open(os.path.join(ROOT, 'tests', fn)).read()
        for m in pat.finditer(src):
            body = m.group(2).strip()
            if re.search(r'\d', body) and body not in allowed and 'compare.' not in body:
                raise AssertionError('%s: tolerance literal %r -- register it in tests/tolerances.py' % (fn, m.group(0)))


def test_design_md_names_every_exception():
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    for key, e in tolerances.EXCEPTIONS.items():
        assert key in text, 'DESIGN.md section 6 does not list the tolerance exception %s' % key
        for v in e['value'].values():
            assert ('%g' % v) in text or repr(v) in text, 'DESIGN.md does not state the value %r of %s' % (v, key)
