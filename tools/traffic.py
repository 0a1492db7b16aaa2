"""HBM traffic and pipe counters per logical step launch from the rocprofv3 passes of tools/prof_workload.sh.

usage: python tools/traffic.py gpurun_out/prof_c4 gpurun_out/prof_c5 gpurun_out/prof_c3 gpurun_out/prof_fwd2048 gpurun_out/prof_c4_both_axes > profiles/r04_traffic.json

FETCH_SIZE and WRITE_SIZE come from separate passes; units KiB per dispatch; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md,
HBM section).  A "logical step launch" = everything one time step of one pass of one batch runs.  `bench.py --steps 1 --warmup 1
--no-e2e` runs 2 fits under the profiler."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_direction as _kd


def kernel_direction(name):
    # the axis-1 pre-pass (blhip_hwide.hpp) runs in both passes: its own row, per launch (two launches per time step of a full fit)
    return 'prepass' if 'hwide_kernel' in name else _kd(name)

FITS = int(os.environ.get('TRAFFIC_FITS', '2'))      # fits under the profiler: bench.py --steps 1 --warmup 1 (fwd2048 steady-state run: 45)
SHAPES = dict(c4=dict(batches=2, T=256, cells=256 * 512 * 512, alg=(16, 32)), c5=dict(batches=4, T=1000, cells=62.5 * 512 * 512, alg=(16, 32)),
              c3=dict(batches=1, T=2000, cells=1024 * 1024, alg=(16, 32)), fwd2048=dict(batches=1, T=200, cells=2048 * 2048, alg=(16, 32)),
              c4_both_axes=dict(batches=2, T=256, cells=256 * 512 * 512, alg=(16, 32)), c4_rows1024=dict(batches=1, T=128, cells=128 * 1024 * 512, alg=(16, 32)),
              coal_hyper1000=dict(batches=1, T=110, cells=256 * 1000, alg=(16, 32)), coal_breakpoints=dict(batches=6, T=41, cells=23400 / 6 * 1000, alg=(16, 32)))


def counters(sub):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            d = kernel_direction(row['Kernel_Name'])
            if d:
                tot[d][row['Counter_Name']] += float(row['Counter_Value'])
                tot[d].setdefault('_kernels', set()).add(row['Kernel_Name'].split('(')[0].replace('void ', ''))
    return tot


def stats(out):
    res = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            d = kernel_direction(row['Name'])
            if d:
                res[d][0] += int(row['Calls']); res[d][1] += float(row['TotalDurationNs'])
    return res


out = {}
for path in sys.argv[1:]:
    w = os.path.basename(path.rstrip('/')).replace('prof_', '')
    sh = SHAPES[w]
    steps = FITS * sh['batches'] * sh['T']
    cs = collections.defaultdict(dict)
    for sub in sorted(glob.glob(os.path.join(path, 'pmc*'))):
        if os.path.isdir(sub):
            for d, c in counters(sub).items():
                cs[d].update(c)
    st = stats(path)
    res = {}
    for d, alg in (('forward', sh['alg'][0]), ('backward', sh['alg'][1]), ('prepass', 16)):
        c = cs.get(d)
        if not c:
            continue
        fb, wb = c.get('FETCH_SIZE', 0.0) * 1024 * 2, c.get('WRITE_SIZE', 0.0) * 1024
        r = dict(hbm_bytes_per_step_launch=(fb + wb) / steps, fetch_bytes_per_step_launch=fb / steps, write_bytes_per_step_launch=wb / steps,
                 hbm_bytes_per_cell_step=(fb + wb) / steps / sh['cells'], streaming_formulation_bytes_per_cell_step=alg,
                 kernel_ns_per_step_launch=st[d][1] / steps if d in st else None, kernel_launches=st[d][0] if d in st else None,
                 kernels=sorted(c.get('_kernels', [])))
        if 'SQ_WAVE_CYCLES' in c:
            wc = c['SQ_WAVE_CYCLES']
            r['sq'] = dict(valu_instructions_per_step_launch=c.get('SQ_INSTS_VALU', 0) / steps, mfma_instructions_per_step_launch=c.get('SQ_INSTS_MFMA', 0) / steps,
                           valu_active_fraction_of_wave_time=c.get('SQ_ACTIVE_INST_VALU', 0) / wc, wait_any_fraction=c.get('SQ_WAIT_ANY', 0) / wc,
                           wait_inst_any_fraction=c.get('SQ_WAIT_INST_ANY', 0) / wc, active_inst_any_fraction=c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                           mfma_busy_cycles_per_step_launch=c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / steps,
                           note='SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; 2 waves per SIMD')
        res[d] = r
    out[w] = res
out['source'] = ('tools/prof_workload.sh <w>: rocprofv3 --kernel-trace --stats, then separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters) on '
                 '`python bench.py --workload <w> --steps 1 --warmup 1 --no-extra --no-cpu --no-pmc --no-e2e`; FETCH_SIZE x 2 (gfx950); tools/traffic.py')
json.dump(out, sys.stdout, indent=1, default=list)
