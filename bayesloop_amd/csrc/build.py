"""Builds libblhip.so in-tree for gfx950:  python -m bayesloop_amd.csrc.build [--force]"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'libblhip.so')


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


def slices():
    """The translation units of the library: blhip.hip (C-ABI, host orchestration, every kernel family but one) and the slices of the
    chain-resident kernels (blhip_chain_tu.hip -DBLC_TU=k, k = 1 .. blcl::N_SLICES of blhip_chain_launch.hpp) -- a few hundred template
    instantiations that were three quarters of a 4-minute single-unit build."""
    import re
    n = int(re.search(r'constexpr int N_SLICES = (\d+);', open(os.path.join(HERE, 'blhip_chain_launch.hpp')).read()).group(1))
    return [('blhip', 'blhip.hip', [])] + [('chain_tu%d' % k, 'blhip_chain_tu.hip', ['-DBLC_TU=%d' % k]) for k in range(1, n + 1)]


def compile_and_link(out, objdir, flags=(), force=False, verbose=True):
    """Objects in parallel (one hipcc per unit, as many at a time as there are cores), then one link.  An object is reused when it is
    newer than every source / header (a change of blhip.hip alone recompiles one unit)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    newest = max(os.path.getmtime(d) for d in deps() if os.path.exists(d))
    jobs, objs = [], []
    for name, src, defs in slices():
        obj = os.path.join(objdir, name + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            jobs.append([hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj] + list(defs) + list(flags))

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)

    if jobs:
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
            list(pool.map(run, jobs))
    # librccl is NOT linked: the communicator entry points (blhip_comm_*) dlopen it on first use
    run([hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + ['-ldl'])
    for junk in glob.glob(out + '.*'):      # offload-bundle side files some hipcc versions leave behind
        try:
            os.remove(junk)
        except OSError:
            pass
    return out


def build_variant(name, flags, verbose=True):
    """Development variants of the library (e.g. -DBLR_PROF: phase stamps of the time-resident kernel), built next to the product
    as libblhip_<name>.so and selected with BLHIP_LIBRARY=...; never the default."""
    out = os.path.join(os.path.dirname(HERE), 'libblhip_%s.so' % name)
    return compile_and_link(out, os.path.join(HERE, '_obj', name), flags, force=True, verbose=verbose)


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    return compile_and_link(OUT, os.path.join(HERE, '_obj', 'product'), (), force=force, verbose=verbose)


if __name__ == '__main__':
    if '--variant' in sys.argv:
        k = sys.argv.index('--variant')
        build_variant(sys.argv[k + 1], sys.argv[k + 2:])
    else:
        build(force='--force' in sys.argv)
