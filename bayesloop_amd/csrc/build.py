"""Builds libblhip.so in-tree for gfx950:  python -m bayesloop_amd.csrc.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'libblhip.so')
SOURCES = ['blhip.hip']
DEPS = ['blhip.hip', 'blhip_kernels.hpp', 'blhip_fast.hpp', 'blhip_persist1d.hpp', os.path.join('..', '..', 'include', 'blhip.h')]


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in DEPS if os.path.exists(os.path.join(HERE, d)))


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', OUT] + SOURCES
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
