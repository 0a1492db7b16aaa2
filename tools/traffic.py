"""HBM traffic per logical step launch from the rocprofv3 PMC passes of tools/prof.sh.

usage: python tools/traffic.py gpurun_out/prof_c4 [gpurun_out/prof_fwd2048] > profiles/r01_traffic.json

FETCH_SIZE and WRITE_SIZE are collected in separate passes (they do not fit one pass).  Units: KiB per dispatch;
FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section: the counter tallies 128-B requests at 64 B).
A "logical step launch" = all radius-bucket launches of one time step of the batch (what bench.py times with HIP events),
so the per-kernel sums are divided by the number of time steps x batches that ran.
"""
import csv, glob, json, os, sys, collections

def load(sub, counter):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(os.path.join(sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == counter:
                tot[row['Kernel_Name']] += float(row['Counter_Value']); cnt[row['Kernel_Name']] += 1
    return tot, cnt

def stats(out):
    res = {}
    for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            res[row['Name']] = (int(row['Calls']), float(row['TotalDurationNs']))
    return res

def direction(name):
    # template argument 2 of the step kernels: 0 = forward, 1 = backward
    if 'step_kernel<' not in name: return None
    args = name.split('step_kernel<')[1].split('>')[0].split(',')
    return 'fwd' if args[1].strip() == '0' else 'bwd'

def summarise(out, steps, cells, bytes_fwd=16, bytes_bwd=32):
    fetch, nf = load(os.path.join(out, 'pmc1'), 'FETCH_SIZE')
    write, nw = load(os.path.join(out, 'pmc2'), 'WRITE_SIZE')
    st = stats(out)
    res = {}
    for d, bpc in (('fwd', bytes_fwd), ('bwd', bytes_bwd)):
        names = [k for k in set(fetch) | set(write) if direction(k) == d]
        if not names: continue
        fb = sum(fetch.get(k, 0.0) for k in names) * 1024 * 2
        wb = sum(write.get(k, 0.0) for k in names) * 1024
        ns = sum(st[k][1] for k in st if direction(k) == d)
        calls = sum(st[k][0] for k in st if direction(k) == d)
        res[d] = dict(hbm_bytes_per_step_launch=(fb + wb) / steps, fetch_bytes_per_step_launch=fb / steps,
                      write_bytes_per_step_launch=wb / steps, algorithmic_bytes_per_step_launch=bpc * cells,
                      ratio=(fb + wb) / steps / (bpc * cells), kernel_ns_per_step_launch=ns / steps, kernel_launches=calls,
                      kernels=sorted(set(k.split('(')[0].replace('void ', '') for k in names)))
    return res

if __name__ == '__main__':
    out = {}
    c4 = sys.argv[1]
    # bench.py --steps 1 --warmup 1 => 2 fits x 2 batches of 256 chains x T = 256 time steps per direction
    out.update(summarise(c4, steps=2 * 2 * 256, cells=256 * 512 * 512))
    if len(sys.argv) > 2:
        r = summarise(sys.argv[2], steps=2 * 200, cells=2048 * 2048)
        if 'fwd' in r:
            out['fwd2048'] = dict(hbm_bytes_per_launch=r['fwd']['hbm_bytes_per_step_launch'], algorithmic_bytes=16 * 2048 * 2048,
                                  ratio=r['fwd']['ratio'], kernel_ns_per_launch=r['fwd']['kernel_ns_per_step_launch'],
                                  kernels=r['fwd']['kernels'])
    out['source'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof.sh) on `python bench.py --steps 1 '
                     '--warmup 1 --no-extra --no-cpu` (C4) and `--workload fwd2048`; FETCH_SIZE x2 (gfx950 correction); tools/traffic.py')
    json.dump(out, sys.stdout, indent=1)
