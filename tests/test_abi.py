"""The C-ABI library loads here (no GPU needed) and exports every symbol include/blhip.h declares."""
import ctypes
import os
import re

from bayesloop_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'blhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(blhip_[a-z_]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_abi.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_abi.library_path())
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert _abi.load().blhip_abi_version() == _abi.ABI_VERSION


def test_struct_layout_matches_header():
    # sizes implied by the field lists of include/blhip.h on LP64
    assert ctypes.sizeof(_abi.Op) == 16
    assert ctypes.sizeof(_abi.Problem) == 4 + 4 + 3 * 8 * _abi.MAX_DIM + 8 + 4 + 4 + 6 * 8 + 8 + 8 + 8 + 4 + 4 + 8 + 8      # n, marginal, lattice: [BLHIP_MAX_DIM]; ... backward_init, prior_token
    assert ctypes.sizeof(_abi.Result) == 5 * 8
    assert ctypes.sizeof(_abi.Timing) == 4 * 8 + 5 * 8 + 2 * 4 + 4 * 8 + 6 * 4


def test_no_gpu_fails_loudly():
    import pytest
    from bayesloop_amd.engine import HipEngine
    from bayesloop_amd.exceptions import BackendError
    if _abi.load().blhip_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(BackendError):
        HipEngine(0)


def test_header_is_plain_c_and_the_c_demo_links(tmp_path):
    """include/blhip.h compiles as C99 (no C++ / torch types) and examples/c_abi_demo.c links against libblhip.so; without
    a GPU the demo prints the ABI version and exits with 0 (its compute part is exercised by the -m gpu test)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        import pytest
        pytest.skip('gcc not available')
    exe = str(tmp_path / 'c_abi_demo')
    libdir = os.path.dirname(_abi.library_path())
    cmd = ['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-O2', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_abi_demo.c'), '-o', exe, '-L' + libdir, '-lblhip', '-Wl,-rpath,' + libdir, '-lm']
    subprocess.run(cmd, check=True, capture_output=True)
    if _abi.load().blhip_device_count() > 0:
        return
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and 'ABI version %d' % _abi.ABI_VERSION in out.stdout


def test_integration_md_ctypes_stub_matches_the_binding():
    """The reference-side ctypes stub INTEGRATION.md shows a maintainer: its structure definitions, exec'ed as written, must have the
    sizes and field offsets of bayesloop_amd/_abi.py (i.e. of include/blhip.h)."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = text.split('```python')[1].split('```')[0]
    struct_src = block.split('lib.blhip_create.restype')[0]
    struct_src = struct_src.replace("lib = C.CDLL('libblhip.so')", 'lib = None')       # (definitions only: nothing is loaded here)
    ns = {}
    exec(struct_src, ns)
    for name in ('Op', 'Problem', 'Result'):
        doc, ours = ns[name], getattr(_abi, name)
        assert ctypes.sizeof(doc) == ctypes.sizeof(ours), name
        assert [f[0] for f in doc._fields_] == [f[0] for f in ours._fields_], name
        for field in (f[0] for f in ours._fields_):
            assert getattr(doc, field).offset == getattr(ours, field).offset, (name, field)
            assert getattr(doc, field).size == getattr(ours, field).size, (name, field)
    # the rest of the stub at least parses
    compile(block, 'INTEGRATION.md', 'exec')
