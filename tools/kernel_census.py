"""Which kernel instantiations of libblhip.so did a set of runs launch?  (prune by evidence.)

    python tools/kernel_census.py <kernel_stats.csv | kernel_trace.csv> ...   > profiles/rNN_kernel_census.txt

Input: rocprofv3 --kernel-trace [--stats] CSVs of the runs (the -m gpu suite, bench.py with all extras).  Output: per kernel family the
instantiations the library holds (host stubs: nm -C), how many of them were launched, and the template arguments never seen."""
import collections, csv, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, 'bayesloop_amd', 'libblhip.so')


def norm(name):
    name = re.sub(r'^void ', '', name.strip().strip('"'))
    name = re.sub(r'\(.*$', '', name)                   # argument list
    name = name.replace('(anonymous namespace)::', '')
    return re.sub(r'\s+', '', name)


have = collections.defaultdict(set)
for line in subprocess.run(['nm', '-C', lib], capture_output=True, text=True).stdout.split('\n'):
    if '__device_stub__' in line:
        full = norm(line.split('__device_stub__', 1)[1])
        # the namespace sits in front of __device_stub__: "blc::__device_stub__chain_kernel<...>"
        ns = re.search(r'(\w+)::$', line.split('__device_stub__', 1)[0])
        full = (ns.group(1) + '::' if ns else '') + full
        have[re.sub(r'<.*', '', full)].add(full)
seen = collections.Counter()
for f in sys.argv[1:]:
    for row in csv.DictReader(open(f)):
        name = row.get('Name') or row.get('Kernel_Name') or ''
        if name:
            seen[norm(name)] += int(row.get('Calls') or 1)
print('library: %d kernel instantiations in %d families, %.1f MB' % (sum(len(v) for v in have.values()), len(have), os.path.getsize(lib) / 1e6))
print('runs: %d distinct kernels launched (%d launches)\n' % (len(seen), sum(seen.values())))
for fam in sorted(have, key=lambda k: -len(have[k])):
    inst = have[fam]
    hit = {k for k in inst if k in seen}
    print('%-34s %5d instantiations, %5d launched' % (fam, len(inst), len(hit)))
    if len(inst) > 4:
        # per template-argument position: the values never seen in a launched instantiation
        def args(k):
            m = re.search(r'<(.*)>$', k)
            return [a.strip() for a in m.group(1).split(',')] if m else []
        npos = max(len(args(k)) for k in inst)
        for pos in range(npos):
            allv = sorted({args(k)[pos] for k in inst if len(args(k)) > pos}, key=lambda v: (len(v), v))
            hitv = {args(k)[pos] for k in hit if len(args(k)) > pos}
            missing = [v for v in allv if v not in hitv]
            print('      argument %d: %d values, never launched: %s' % (pos, len(allv), ', '.join(missing) if missing else '-'))
unknown = [k for k in seen if not any(k in v for v in have.values())]
if unknown:
    print('\nlaunched but not matched to a stub (name formatting): %d, e.g. %s' % (len(unknown), unknown[:3]))
