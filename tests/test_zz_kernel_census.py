"""The kernels the library ships are the kernels the tests have run.

Every __global__ instantiation of libblhip.so registers itself when the library is loaded (bayesloop_amd/csrc/blhip_err.hpp: blreg) and
counts its launches; blhip_kernel_census (include/blhip.h) reports the list.  On CPU: the registry is complete (one entry per device
stub of the binary).  On the GPU, as the LAST test of the -m gpu suite (this file sorts behind the others): no entry with zero launches --
an instantiation nobody compared with the oracle does not ship.  tests/test_kernel_sweep.py is what walks the product space."""
import os
import re
import subprocess

import pytest

from bayesloop_amd import _abi
from conftest import kernel_census

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_lists_every_kernel_of_the_binary():
    rows = kernel_census()
    names = [n for _, n in rows]
    assert len(names) == len(set(names)) and len(names) > 100
    nm = subprocess.run(['nm', '-C', _abi.library_path()], capture_output=True, text=True).stdout
    stubs = [l for l in nm.split('\n') if '__device_stub__' in l]
    # (one host stub per kernel instantiation: a kernel launched around the registry -- a bare hipLaunchKernelGGL -- would show up here)
    assert len(stubs) == len(names), (len(stubs), len(names))
    for fam in ('blc::chain_kernel<', 'blc::chain_fold2_kernel<', 'blr::resident_kernel<', 'blm::mfma_step_kernel<', 'blk::reduce_partials_kernel'):
        assert any(n.startswith(fam) for n in names), fam


def test_no_launch_site_bypasses_the_registry():
    src = os.path.join(ROOT, 'bayesloop_amd', 'csrc')
    for fn in sorted(os.listdir(src)):
        if not fn.endswith(('.hip', '.hpp')) or fn == 'blhip_err.hpp':
            continue
        text = open(os.path.join(src, fn)).read()
        text = re.sub(r'//[^\n]*', '', text)
        for m in re.finditer(r'hipLaunchKernelGGL\(|<<<', text):
            line = text[:m.start()].count('\n') + 1
            ctx = text[max(0, m.start() - 200):m.start()]
            if text[m.end():m.end() + 5] == 'KERN,':        # (blhip_chain_tu.hip: launch_chain_ptr -- its callers count, see the launch_chain_fn macro)
                continue
            assert 'blreg::hit<' in ctx, '%s:%d launches a kernel the registry does not count (use BL_LAUNCH)' % (fn, line)


@pytest.mark.gpu
def test_every_kernel_the_library_holds_was_launched_in_this_session(request):
    """Runs last.  Only meaningful for the whole -m gpu suite (a -k selection or a single file skips it; BLHIP_CENSUS_STRICT=1 forces it)."""
    cfg = request.config
    whole = not cfg.option.keyword and all(os.path.isdir(a.split('::')[0]) for a in cfg.args)
    if os.environ.get('BLHIP_CENSUS_STRICT', '') == '0' or (not whole and os.environ.get('BLHIP_CENSUS_STRICT', '') != '1'):
        pytest.skip('not the whole suite')
    rows = kernel_census()
    never = [n for c, n in rows if c == 0]
    assert not never, '%d of %d kernel instantiations were never launched (full list: gpurun_out/kernel_census.txt), e.g.\n  %s' % (
        len(never), len(rows), '\n  '.join(never[:40]))
