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


def test_the_slices_of_the_chain_kernels_cover_every_declared_launcher():
    """build.py compiles blhip_chain_tu.hip once per slice (-DBLC_TU=k, k = 1 .. blcl::N_SLICES): every launcher the header declares is
    defined in exactly one slice, and the slice count the build reads is the header's."""
    import re
    from bayesloop_amd.csrc import build
    here = os.path.join(ROOT, 'bayesloop_amd', 'csrc')
    hdr = open(os.path.join(here, 'blhip_chain_launch.hpp')).read()
    tu = open(os.path.join(here, 'blhip_chain_tu.hip')).read()
    n = int(re.search(r'constexpr int N_SLICES = (\d+);', hdr).group(1))
    units = build.slices()
    assert len(units) == n + 1 and units[0][1] == 'blhip.hip'
    assert [u[2] for u in units[1:]] == [['-DBLC_TU=%d' % k] for k in range(1, n + 1)]
    declared = re.findall(r'^void (\w+)\(hipStream_t', hdr, flags=re.M)
    assert len(declared) == len(set(declared)) and len(declared) >= n
    body = tu[tu.index('namespace blcl {'):]
    sections = re.split(r'#(?:el)?if BLC_TU == (\d+)', body)          # [pre, k1, text1, k2, text2, ...]
    defined = {}
    for k, text in zip(sections[1::2], sections[2::2]):
        for name in re.findall(r'^void (\w+)\(hipStream_t', text, flags=re.M):
            assert name not in defined, '%s defined in slices %s and %s' % (name, defined[name], k)
            defined[name] = int(k)
    assert set(defined) == set(declared), (set(declared) ^ set(defined))
    assert set(defined.values()) == set(range(1, n + 1))               # (no empty slice: every compilation earns its place)


def test_objects_are_keyed_on_the_toolchain_and_written_atomically(tmp_path, monkeypatch):
    """Advisor finding (round 4): objects were reused on mtime alone.  A fake hipcc records its calls: a second build with the same
    compiler reuses every object, another compiler (other --version text) or other flags recompile all of them, every output is written
    under a temporary name first, and an interrupted compile leaves no object behind."""
    import stat
    from bayesloop_amd.csrc import build
    log = tmp_path / 'calls.log'
    fake = tmp_path / 'hipcc'
    fake.write_text('#!/bin/sh\n'
                    'if [ "$1" = "--version" ]; then echo "fake hipcc $FAKE_VER"; exit 0; fi\n'
                    'echo "$@" >> %s\n'
                    'out=""; prev=""; for a in "$@"; do if [ "$prev" = "-o" ]; then out="$a"; fi; prev="$a"; done\n'
                    'case "$out" in *.tmp*) ;; *) echo "NOT-TEMPORARY $out" >> %s;; esac\n'
                    'if [ -n "$FAKE_FAIL" ]; then echo partial > "$out"; exit 1; fi\n'
                    'echo obj > "$out"\n' % (log, log))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv('HIPCC', str(fake))
    monkeypatch.setenv('FAKE_VER', '1')
    objdir, out = str(tmp_path / 'obj'), str(tmp_path / 'lib.so')
    n_units = len(build.slices())

    def calls():
        return [l for l in log.read_text().splitlines() if ' -c ' in l] if log.exists() else []
    build.compile_and_link(out, objdir, verbose=False)
    assert len(calls()) == n_units and os.path.exists(out)
    build.compile_and_link(out, objdir, verbose=False)
    assert len(calls()) == n_units                                  # same toolchain, nothing newer: every object reused
    monkeypatch.setenv('FAKE_VER', '2')                             # another compiler
    build.compile_and_link(out, objdir, verbose=False)
    assert len(calls()) == 2 * n_units
    build.compile_and_link(out, objdir, flags=['-DX=1'], verbose=False)   # other flags
    assert len(calls()) == 3 * n_units
    assert 'NOT-TEMPORARY' not in log.read_text()
    # an interrupted compile: no object under its final name, no stamp
    monkeypatch.setenv('FAKE_FAIL', '1')
    with pytest.raises(subprocess.CalledProcessError):
        build.compile_and_link(out, objdir, force=True, flags=['-DX=2'], verbose=False)
    assert not os.path.exists(os.path.join(objdir, 'STAMP'))
    assert not [f for f in os.listdir(objdir) if '.tmp' in f]
