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


BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def toolchain_key(flags=()):
    """What an object depends on beside the sources: the compiler (its --version text: a ROCm upgrade or another HIPCC changes it) and the
    flags.  Kept in <objdir>/STAMP; objects built under another key are never reused."""
    import hashlib
    try:
        ver = subprocess.run([hipcc(), '--version'], capture_output=True, text=True, timeout=60).stdout
    except Exception:           # noqa: BLE001 -- no compiler: the key still separates flag sets
        ver = 'unknown'
    return hashlib.sha256((hipcc() + '\n' + ver + '\n' + ' '.join(BASE_FLAGS + list(flags))).encode()).hexdigest()[:16]


class _BuildLock:
    """One build of an object directory at a time (pytest -n 4, N ranks starting with a stale library): an exclusive flock on
    <objdir>/LOCK held for the whole compile + link."""
    def __init__(self, objdir):
        self.path = os.path.join(objdir, 'LOCK')

    def __enter__(self):
        import fcntl
        self.f = open(self.path, 'w')
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def unit_newest(depfile, fallback):
    """Newest modification time among the files a unit's last compilation read (make-style dependency file of hipcc -MD; system headers
    outside this tree are skipped); `fallback` when there is no such file or one of its entries has disappeared."""
    try:
        text = open(depfile).read()
    except OSError:
        return fallback
    newest = 0.0
    for tok in text.replace('\\\n', ' ').split():
        if tok.endswith(':') or tok.startswith(('/opt/', '/usr/')):
            continue
        path = tok if os.path.isabs(tok) else os.path.join(HERE, tok)
        if not os.path.exists(path):
            return fallback
        newest = max(newest, os.path.getmtime(path))
    return newest or fallback


def compile_and_link(out, objdir, flags=(), force=False, verbose=True):
    """Objects in parallel (one hipcc per unit, as many at a time as there are cores), then one link.  An object is reused when it is
    newer than every source / header AND was built by the same compiler with the same flags (STAMP); every output (objects, the
    library) is written to a temporary name and renamed into place, so an interrupted hipcc leaves nothing that looks finished."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    with _BuildLock(objdir):
        key = toolchain_key(flags)
        stamp = os.path.join(objdir, 'STAMP')
        same_tools = os.path.exists(stamp) and open(stamp).read().strip() == key
        newest_all = max(os.path.getmtime(d) for d in deps() if os.path.exists(d))
        jobs, objs = [], []
        for name, src, defs in slices():
            obj = os.path.join(objdir, name + '.o')
            objs.append(obj)
            # an object depends on the files its own compilation read (the compiler's dependency file, written beside it: a header only
            # some units include recompiles those units only); no dependency file: on every source / header
            newest = unit_newest(obj + '.d', newest_all)
            if force or not same_tools or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
                jobs.append((obj, [hipcc()] + BASE_FLAGS + ['-c', src, '-MD', '-MF', obj + '.d'] + list(defs) + list(flags)))

        def run(job):
            target, cmd = job
            tmp = '%s.tmp%d' % (target, os.getpid())
            if verbose:
                print(' '.join(cmd + ['-o', target]), flush=True)
            try:
                dep = None
                if '-MF' in cmd:                        # the dependency file is renamed into place with its object
                    cmd = list(cmd)
                    dep = cmd[cmd.index('-MF') + 1]
                    cmd[cmd.index('-MF') + 1] = tmp + '.d'
                subprocess.check_call(cmd + ['-o', tmp], cwd=HERE)
                if dep and os.path.exists(tmp + '.d'):
                    os.replace(tmp + '.d', dep)
                os.replace(tmp, target)
            finally:
                for junk in glob.glob(tmp + '*'):
                    try:
                        os.remove(junk)
                    except OSError:
                        pass

        if jobs:
            if not same_tools and os.path.exists(stamp):
                os.remove(stamp)                    # (objects of two toolchains are never mixed, also after an interrupted build)
            with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
                list(pool.map(run, jobs))
            with open(stamp + '.tmp', 'w') as f:
                f.write(key + '\n')
            os.replace(stamp + '.tmp', stamp)
        # librccl is NOT linked: the communicator entry points (blhip_comm_*) dlopen it on first use
        run((out, [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-ldl']))
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


LAST_ACTION = [None]        # 'reused' / 'compiled': what the last build() call did (__graft_entry__.build prints it)


def build(force=False, verbose=True):
    """-> path of libblhip.so.  The binary is git-ignored and travels with a push of the working tree: where it is present and newer than
    every source this is a no-op ('reused'); on a fresh clone it is a ~2-minute compile of 23 translation units ('compiled')."""
    if not force and not stale():
        LAST_ACTION[0] = 'reused'
        return OUT
    LAST_ACTION[0] = 'compiled'
    return compile_and_link(OUT, os.path.join(HERE, '_obj', 'product'), (), force=force, verbose=verbose)


if __name__ == '__main__':
    if '--variant' in sys.argv:
        k = sys.argv.index('--variant')
        build_variant(sys.argv[k + 1], sys.argv[k + 2:])
    else:
        build(force='--force' in sys.argv)
