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
    assert ctypes.sizeof(_abi.Problem) == 4 + 4 + 16 + 16 + 16 + 8 + 4 + 4 + 6 * 8 + 8 + 8 + 8 + 4 + 4
    assert ctypes.sizeof(_abi.Result) == 5 * 8
    assert ctypes.sizeof(_abi.Timing) == 4 * 8 + 5 * 8 + 2 * 4


def test_no_gpu_fails_loudly():
    import pytest
    from bayesloop_amd.engine import HipEngine
    from bayesloop_amd.exceptions import BackendError
    if _abi.load().blhip_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(BackendError):
        HipEngine(0)
