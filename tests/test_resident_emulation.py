"""The time-resident kernel's per-thread phase functions (bayesloop_amd/csrc/blhip_resident.hpp) compile on the host too
(-DBLR_EMULATE); tools/emu/resident_emu.cpp runs them sequentially -- all tiles through a phase, then the next phase -- against a dense
evaluation of the same forward / backward recursion: tile / segment / halo / tagged-strip indexing, the gathering waves' shares of the
lagged sum and the lag bookkeeping, without a GPU.  (The hand-off protocol itself -- tags seen before data, re-polls -- asserts inside the
emulated primitives: a strip element consumed before it was published aborts the run.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resident_kernel_phase_functions_against_a_dense_evaluation(tmp_path):
    if shutil.which('g++') is None:
        pytest.skip('g++ not available')
    exe = str(tmp_path / 'resident_emu')
    subprocess.run(['g++', '-O1', '-std=c++17', '-DBLR_EMULATE', '-I', os.path.join(ROOT, 'bayesloop_amd', 'csrc'),
                    os.path.join(ROOT, 'tools', 'emu', 'resident_emu.cpp'), '-o', exe], check=True, capture_output=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('tile ')]
    assert len(lines) >= 10 and all(l.rstrip().endswith('ok (0 mismatches)') for l in lines), out.stdout[-3000:]
