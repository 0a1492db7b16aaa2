"""Build hygiene: the library is rebuilt when ANY of its sources changes, and no build product is tracked by git."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_header_is_a_build_dependency():
    from bayesloop_amd.csrc import build
    deps = {os.path.basename(d) for d in build.deps()}
    here = os.path.join(ROOT, 'bayesloop_amd', 'csrc')
    for f in os.listdir(here):
        if f.endswith(('.hip', '.hpp', '.h')):
            assert f in deps, '%s is not a dependency of libblhip.so' % f
    assert 'blhip.h' in deps                       # the public C header


def test_touching_a_kernel_header_makes_the_library_stale():
    from bayesloop_amd.csrc import build
    if not os.path.exists(build.OUT):
        pytest.skip('library not built')
    hdr = os.path.join(ROOT, 'bayesloop_amd', 'csrc', 'blhip_mfma.hpp')
    st = os.stat(hdr)
    lib_t = os.path.getmtime(build.OUT)
    try:
        os.utime(hdr, (lib_t + 10, lib_t + 10))
        assert build.stale()
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
    # (whether it is stale right now depends on when the tree was last built; __graft_entry__.build() rebuilds if so)


def test_no_build_products_are_tracked():
    try:
        files = subprocess.run(['git', 'ls-files'], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split('\n')
    except Exception:
        pytest.skip('not a git checkout')
    bad = [f for f in files if f.endswith(('.so', '.o', '.a', '.hsaco', '.co')) or '.so.' in f or f.startswith('build_ubench/')]
    for f in files:
        path = os.path.join(ROOT, f)
        if f and os.path.isfile(path) and os.path.getsize(path) > 4 and not f.endswith(('.npz', '.npy', '.png', '.pdf')):
            with open(path, 'rb') as fh:
                if fh.read(4) == b'\x7fELF':
                    bad.append(f)
    assert not bad, 'build products in git: %s' % bad
