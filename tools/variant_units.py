"""A development variant of the library that differs from the product build in a few translation units only:
    python tools/variant_units.py <name> <units> [flags ...]      e.g.  fe0 chain_tu3,chain_tu9 -DBLC_FASTEDGE=0
-> bayesloop_amd/libblhip_<name>.so = the product's objects with <units> recompiled under the extra flags (seconds per unit instead of the
minutes of build.py --variant).  The product objects must be current (python -m bayesloop_amd.csrc.build)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesloop_amd.csrc import build as B

name, units, flags = sys.argv[1], sys.argv[2].split(','), sys.argv[3:]
prod = os.path.join(B.HERE, '_obj', 'product')
objdir = os.path.join(B.HERE, '_obj', 'units_' + name)
os.makedirs(objdir, exist_ok=True)
objs, jobs = [], []
for uname, src, defs in B.slices():
    if uname in units:
        obj = os.path.join(objdir, uname + '.o')
        jobs.append([B.hipcc()] + B.BASE_FLAGS + ['-c', src] + list(defs) + flags + ['-o', obj])
    else:
        obj = os.path.join(prod, uname + '.o')
    objs.append(obj)
assert len(jobs) == len(units), 'unknown unit in %r' % units
with ThreadPoolExecutor(max_workers=8) as pool:
    list(pool.map(lambda c: subprocess.check_call(c, cwd=B.HERE), jobs))
out = os.path.join(ROOT, 'bayesloop_amd', 'libblhip_%s.so' % name)
subprocess.check_call([B.hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-ldl', '-o', out])
print(out)
