"""HBM traffic per logical step launch from the rocprofv3 PMC passes of tools/prof_r02.sh (round 2).

usage: python tools/traffic_r02.py gpurun_out/prof_r02_c4 gpurun_out/prof_r02_fwd2048 gpurun_out/prof_r02_c3 > profiles/r02_traffic.json

FETCH_SIZE and WRITE_SIZE come from separate passes (they do not fit one pass).  Units: KiB per dispatch; FETCH_SIZE is
doubled on gfx950 (MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B).  A "logical step launch" = everything one
time step of one pass runs: all radius-bucket launches of the batch (C4), or 1/T of the one time-resident launch (fwd2048, C3).
"""
import collections, csv, glob, json, os, sys


def load(sub, counter):
    tot = collections.defaultdict(float)
    for f in glob.glob(os.path.join(sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == counter:
                tot[row['Kernel_Name']] += float(row['Counter_Value'])
    return tot


def stats(out):
    res = {}
    for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            res[row['Name']] = (int(row['Calls']), float(row['TotalDurationNs']))
    return res


def busy_ns(out, d):
    """Time at least one kernel of direction d was running (union of the dispatch intervals of the kernel trace): the radius buckets
    of a step run concurrently on separate streams, so the SUM of their durations over-counts the wall time of a logical step."""
    iv = []
    for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if direction(row['Kernel_Name']) == d:
                iv.append((int(row['Start_Timestamp']), int(row['End_Timestamp'])))
    iv.sort()
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def direction(name):
    if 'resident_kernel<' in name:
        return 'bwd' if name.split('resident_kernel<')[1].split('>')[0].split(',')[-1].strip() in ('true', '1') else 'fwd'
    if 'chain_kernel<' in name:          # blc::chain_kernel<NK, NTW, BWD, STORE>
        return 'bwd' if name.split('chain_kernel<')[1].split('>')[0].split(',')[2].strip() in ('true', '1') else 'fwd'
    if 'step_kernel<' in name:
        return 'fwd' if name.split('step_kernel<')[1].split('>')[0].split(',')[1].strip() == '0' else 'bwd'
    return None


def summarise(out, steps, cells, bytes_fwd=16, bytes_bwd=32):
    """steps: logical step launches per direction that ran under the profiler (fits x batches x T)."""
    fetch, write, st = load(os.path.join(out, 'pmc1'), 'FETCH_SIZE'), load(os.path.join(out, 'pmc2'), 'WRITE_SIZE'), stats(out)
    res = {}
    for d, bpc in (('fwd', bytes_fwd), ('bwd', bytes_bwd)):
        names = [k for k in set(fetch) | set(write) if direction(k) == d]
        if not names:
            continue
        fb = sum(fetch.get(k, 0.0) for k in names) * 1024 * 2
        wb = sum(write.get(k, 0.0) for k in names) * 1024
        ns = sum(v[1] for k, v in st.items() if direction(k) == d)
        calls = sum(v[0] for k, v in st.items() if direction(k) == d)
        res[d] = dict(hbm_bytes_per_step_launch=(fb + wb) / steps, fetch_bytes_per_step_launch=fb / steps,
                      write_bytes_per_step_launch=wb / steps, algorithmic_bytes_per_step_launch=bpc * cells,
                      ratio=(fb + wb) / steps / (bpc * cells), kernel_ns_per_step_launch=ns / steps,
                      kernel_busy_ns_per_step_launch=busy_ns(out, d) / steps, kernel_launches=calls,
                      kernels=sorted(set(k.split('(')[0].replace('void ', '') for k in names)))
    return res


if __name__ == '__main__':
    out = {}
    for path in sys.argv[1:]:
        w = os.path.basename(path.rstrip('/')).replace('prof_r02_', '')
        if w == 'c4':        # bench.py --steps 1 --warmup 1 => 3 fits (warm-up, timed, end-to-end) x 2 batches x T = 256 logical launches per direction;
            # cells = the average batch (512 chains / 2) -- the unit bench.py's roofline uses (cells_per_launch = cell-steps of the pass / launches)
            out.update(summarise(path, steps=3 * 2 * 256, cells=256 * 512 * 512))
        elif w == 'fwd2048':  # 3 fits x 200 steps in one resident launch each
            r = summarise(path, steps=3 * 200, cells=2048 * 2048)
            if 'fwd' in r:
                out['fwd2048'] = dict(r['fwd'], hbm_bytes_per_launch=r['fwd']['hbm_bytes_per_step_launch'])
        elif w == 'c5':       # 3 fits x 3 batches x 1000 steps; a logical step launch = all launches of one time step of a batch (88 / 88 / 74 chains)
            r = summarise(path, steps=3 * 3 * 1000, cells=250.0 / 3.0 * 512 * 512)
            out['c5'] = r
        elif w == 'c3':
            r = summarise(path, steps=3 * 2000, cells=1024 * 1024)
            out['c3'] = r
    out['source'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_r02.sh) on `python bench.py --workload <w> '
                     '--steps 1 --warmup 1 --no-extra --no-cpu`; FETCH_SIZE x2 (gfx950 correction); tools/traffic_r02.py.  C4: per logical '
                     'step launch of a batch (average batch = 256 chains); fwd2048 / c3: the time-resident kernel, per time step')
    json.dump(out, sys.stdout, indent=1)
