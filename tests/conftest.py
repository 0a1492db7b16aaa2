import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


if os.environ.get('PYTEST_XDIST_WORKER'):
    # several test processes share ONE GPU (pytest -n 4): a resident launch may wait long for blocks another process's kernels keep off
    # the chip -- the flat 2-second bound of round 3 instead of the pass-scaled default (resident_timeout_s)
    # ... and no co-residency probe (blhip_timing.resident_probe): its 2-ms bound would park the resident paths whenever another worker's
    # kernel is on the chip, and the tests that assert WHICH kernel ran would fail with it
    os.environ.setdefault('BLHIP_ENGINE_OPTS', 'resident_timeout_s=4,resident_probe=0')


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _rearm_resident_paths(request):
    """A resident launch that gives up (four test processes share the GPU: a block can starve) parks the resident paths of its context
    for the next 8 fits -- the product's behaviour -- and every following test that asserts WHICH kernel ran would fail with it.  Each
    GPU test starts from an armed context; the test in which a give-up happens still reports it."""
    if 'gpu' in request.keywords:
        try:
            import bayesloop_amd.engine as em
            eng = em._engine
            if eng is not None and hasattr(eng, 'ctx'):
                eng.set_option('resident_ok', 1)
        except Exception:
            pass
    yield


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: takes more than a few seconds on CPU')


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How many localEvidence entries the parity comparisons of this session held to the 1e-9 bar, and how many to a registered
    looser tolerance (tests/tolerances.py: ILL_LOCAL_EVIDENCE / WIDE_FILTER_2D, steps with denormal likelihood cells)."""
    try:
        import compare
    except Exception:
        return
    c = compare.COUNTS
    if c['local_at_bar'] + c['local_loosened'] + c['local_nan']:
        terminalreporter.write_line('localEvidence entries compared: %d at the 1e-9 bar, %d at a registered looser tolerance (%d of them '
                                    'additionally pinned at the bar through the sum over the cells with a normal likelihood value), '
                                    '%d NaN on both sides (0/0 in the reference)' % (c['local_at_bar'], c['local_loosened'],
                                                                                      c.get('local_partial', 0), c['local_nan']))
    out = os.environ.get('BLHIP_PARITY_COUNTS')
    if out:
        import json
        with open(out, 'w') as f:
            json.dump(c, f)


def kernel_census():
    """[(launches in this process, kernel instantiation)] of libblhip.so (blhip_kernel_census, include/blhip.h)."""
    import ctypes
    from bayesloop_amd import _abi
    lib = _abi.load()
    n = lib.blhip_kernel_census(None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    lib.blhip_kernel_census(buf, n + 1)
    rows = []
    for line in buf.value.decode().splitlines():
        cnt, name = line.split('\t', 1)
        rows.append((int(cnt), name))
    return rows


def pytest_sessionfinish(session, exitstatus):
    """After a session that launched kernels: which instantiations of the library ran, which did not (gpurun_out/kernel_census.txt;
    the round's copy is profiles/rNN_kernel_census.txt)."""
    try:
        rows = kernel_census()
    except Exception:
        return
    if not any(c for c, _ in rows):
        return
    try:
        import collections
        import re
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        fam = collections.OrderedDict()
        for c, name in rows:
            f = fam.setdefault(re.sub(r'<.*', '', name), [0, 0, 0])
            f[0] += 1; f[1] += 1 if c else 0; f[2] += c
        lib = os.path.join(ROOT, 'bayesloop_amd', 'libblhip.so')
        with open(os.path.join(out, 'kernel_census.txt'), 'w') as f:
            f.write('library: %d kernel instantiations in %d families, %.1f MB\n' % (len(rows), len(fam), os.path.getsize(lib) / 1e6))
            f.write('this session: %d launched (%d launches), %d never launched\n\n' % (sum(1 for c, _ in rows if c), sum(c for c, _ in rows), sum(1 for c, _ in rows if not c)))
            for name, (n, hit, launches) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
                f.write('%-36s %5d instantiations, %5d launched, %9d launches\n' % (name, n, hit, launches))
            f.write('\nnever launched:\n')
            for c, name in rows:
                if not c:
                    f.write('  %s\n' % name)
    except Exception:
        pass
