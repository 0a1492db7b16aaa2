"""Builds libblhip.so in-tree for gfx950:  python -m bayesloop_amd.csrc.build [--force]"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'libblhip.so')
SOURCES = ['blhip.hip']


def deps():
    """Everything the library is compiled from: every source / header next to this file plus the public C header."""
    d = sorted(glob.glob(os.path.join(HERE, '*.hip')) + glob.glob(os.path.join(HERE, '*.hpp')) + glob.glob(os.path.join(HERE, '*.h')))
    d.append(os.path.join(HERE, '..', '..', 'include', 'blhip.h'))
    return d


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in deps() if os.path.exists(d))


def build_variant(name, flags, verbose=True):
    """Development variants of the library (e.g. -DBLR_PROF: phase stamps of the time-resident kernel), built next to the product
    as libblhip_<name>.so and selected with BLHIP_LIBRARY=...; never the default."""
    out = os.path.join(os.path.dirname(HERE), 'libblhip_%s.so' % name)
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', out] + list(flags) + SOURCES + ['-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return out


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    # librccl is NOT linked: the communicator entry points (blhip_comm_*) dlopen it on first use
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', OUT] + SOURCES + ['-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    for junk in glob.glob(OUT + '.*'):      # offload-bundle side files some hipcc versions leave behind
        try:
            os.remove(junk)
        except OSError:
            pass
    return OUT


if __name__ == '__main__':
    if '--variant' in sys.argv:
        k = sys.argv.index('--variant')
        build_variant(sys.argv[k + 1], sys.argv[k + 2:])
    else:
        build(force='--force' in sys.argv)
