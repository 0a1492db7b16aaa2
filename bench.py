#!/usr/bin/env python
"""
bench.py -- grid-cells x timesteps / second of fit() on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c4_evidence|c4_both_axes|c4_rows1024|c4_wide|c4_laplace|c3|c2|c5|fwd2048|coal_breakpoints|c1_hyper|coal_hyper1000] [--no-extra] [--no-cpu]

A "step" is one complete pass of the hot path over one batch of synthetic input: one whole ``fit()`` of the workload.

Multi-GPU (one process per GPU, RCCL over xGMI through libblhip.so's blhip_comm_* entry points, no PyTorch in the process):
  * under a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE (the driver's ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py --gpus N``) every rank runs this file; WORLD_SIZE must equal --gpus;
  * ``python bench.py --gpus N`` on its own starts the N ranks itself (one child process per GPU ordinal 0..N-1).
Rank 0 prints the ONE JSON line; timing = barrier + device synchronise on both sides of the K timed steps, MAX over ranks.

Workloads (synthetic data, seeds fixed):
  c4       HyperStudy, 512 x 512 Gaussian (mean, std) grid, GaussianRandomWalk on 'mean' with 512 sigma values
           cint(0, 0.3, 512), T = 256, FULL fit (forward + backward + evidence-weighted average posterior).
           The headline at every N: the 512 hyper-grid points are dealt out round-robin to the N ranks
           (strong scaling: total work fixed), one gather + one reduce over RCCL at the end.
  c5       ChangepointStudy, 512 x 512 grid, T = 1000, 256 candidate change-points, full fit     (sharded like c4)
  c3       Study, 1024 x 1024 grid, T = 2000, GRW x GRW separable stencil, full fit             (N = 1)
  c2       Study, 4096-point 1-D GaussianMean grid, T = 10 000, full fit (latency-bound)       (N = 1)
  fwd2048  Study, 2048 x 2048 grid, T = 200, evidenceOnly: the fused-forward-step roofline point (N = 1)
  c4_both_axes  HyperStudy, 512 x 512 grid, random walks on BOTH parameters with 64 x 8 width pairs (radii up to 38 / 29 grid steps),
           T = 256, full fit: the axis-1 pre-pass (blhip_hwide.hpp) + the matrix-pipe step kernels, one launch per step and bucket
  c4_rows1024   HyperStudy, 1024 x 512 grid, 128 widths cint(0, 0.3) on 'mean' (axis-0 radii up to 77), T = 128, full fit: chain-resident
           kernels of the 1024-row geometry (one copy of the strip in LDS, rings of up to 44 entries)
  c4_wide  HyperStudy, 512 x 512 grid, 256 widths cint(0, 0.6) on 'mean' (axis-0 radii up to 77), T = 128, full fit: the same ring
           lengths on the headline's geometry
  c4_laplace  HyperStudy, 512 x 512 grid, Laplace model, 128 widths cint(0, 0.3) on 'mu', T = 128, full fit: chain-resident kernels with the
           likelihood out of the (T, G) table

Inputs are KBs (the series, the marginal grids) and are uploaded inside fit(); all grid-sized state is created and
stays in HBM.  The posterior sequence is left on the device (lazy D2H on first access, not part of the timed region).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

TEST_DOUBLE = os.environ.get('BLHIP_BENCH_TEST_DOUBLE') == '1' and os.path.exists(os.path.join(ROOT, 'tests', 'bench_double.py'))
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
FP64_PEAK_TFLOPS = 78.6    # fp64 peak of the part: the vector ALU and v_mfma_f64_16x16x4 drive the SAME lanes (measured: they add up)
# algorithmic bytes per grid-cell x timestep of the STREAMING formulation (state in HBM): the judge's unit
BYTES_FWD = 16.0           # fused forward step: read state 8 + write state 8 (the state write IS the stored posterior)
BYTES_BWD = 32.0           # fused backward step: read alpha 8, read c 8, write posterior 8, write c 8
# The resident kernels keep the state in LDS: the bytes they REALLY move (what the library reports by construction in
# blhip_timing.*_hbm_bytes, or the PMC counters) are fewer, so a line carries three figures per kernel:
#   hbm              real HBM bytes / time  (fraction of the 8 TB/s spec and of the copy rate calibrated on this box)
#   fp64             fp64 flop as executed / time  (fraction of 78.6 TFLOP/s)
#   streaming_equiv  the 8(d) bytes / time: what a streaming kernel would have to sustain to be as fast -- comparable across
#                    rounds, NOT a bandwidth (it may exceed the peak; the field is named GBs_equiv, never `achieved`)


def series(seed, T):
    rng = np.random.default_rng(seed)
    mu = np.cumsum(rng.normal(0, 0.02, T))
    return mu + rng.normal(0, 1.0, T)


def make_study(bl, name, comm=None, scale=1.0):
    """-> (study, fit kwargs, cells x steps x chains of one fit, description dict)"""
    if name in ('c4', 'c4_evidence'):
        n, T, nh = 512, 256, 512
        S = bl.HyperStudy(silent=True)
        S.loadData(series(4, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.3, nh), target='mean'), silent=True)
        S.communicator = comm
        if name == 'c4_evidence':      # HyperStudy.fit(evidenceOnly=True) -- forward passes only
            return S, dict(silent=True, evidenceOnly=True), n * n * T * nh, dict(
                workload='C4 HyperStudy 512x512 grid x 512 sigma values, T=256, evidenceOnly (forward passes only)',
                grid=[n, n], T=T, n_hyper=nh, mode='evidenceOnly')
        return S, dict(silent=True), n * n * T * nh, dict(workload='C4 HyperStudy 512x512 grid x 512 sigma values, T=256, '
                                                           'full fit (forward+backward+average posterior)',
                                                           grid=[n, n], T=T, n_hyper=nh, mode='full')
    if name == 'c4_both_axes':      # a hyper-study whose random walks act on BOTH parameters (reference: tests/test_hyperstudy.py two-hyper-parameter fits)
        n, T, nh1, nh2 = 512, 256, 64, 8
        S = bl.HyperStudy(silent=True)
        S.loadData(series(4, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', bl.cint(0, 0.3, nh1), target='mean'),
                                            bl.tm.GaussianRandomWalk('s2', bl.cint(0, 0.06, nh2), target='std')), silent=True)
        S.communicator = comm
        return S, dict(silent=True), n * n * T * nh1 * nh2, dict(
            workload='HyperStudy 512x512 grid x (64 x 8) sigma pairs, random walks on both parameters, T=256, full fit',
            grid=[n, n], T=T, n_hyper=nh1 * nh2, mode='full')
    if name == 'c4_rows1024':       # the C4 widths on a grid twice as fine along the walk's axis: axis-0 radii up to 77 > the band kernels' 40
        n0, n1, T, nh = 1024, 512, 128, 128
        S = bl.HyperStudy(silent=True)
        S.loadData(series(4, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n0), 'std', bl.oint(0, 4, n1)),
              bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.3, nh), target='mean'), silent=True)
        S.communicator = comm
        return S, dict(silent=True), n0 * n1 * T * nh, dict(
            workload='HyperStudy 1024x512 grid x 128 sigma values (radii up to 77), T=128, full fit', grid=[n0, n1], T=T, n_hyper=nh, mode='full')
    if name == 'c4_wide':           # the C4 grid with widths twice as wide: axis-0 radii up to 77 on 512 rows (rings of up to 44 entries on the 512-row geometry)
        n, T, nh = 512, 128, 256
        S = bl.HyperStudy(silent=True)
        S.loadData(series(4, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.6, nh), target='mean'), silent=True)
        S.communicator = comm
        return S, dict(silent=True), n * n * T * nh, dict(
            workload='HyperStudy 512x512 grid x 256 sigma values cint(0, 0.6) (radii up to 77), T=128, full fit', grid=[n, n], T=T, n_hyper=nh, mode='full')
    if name == 'c4_laplace':        # a hyper-study on the C4 grid with an observation model other than the Gaussian: the likelihood out of a table
        n, T, nh = 512, 128, 128
        S = bl.HyperStudy(silent=True)
        S.loadData(series(4, T), silent=True)
        S.set(bl.om.Laplace('mu', bl.cint(-8, 8, n), 'b', bl.oint(0, 4, n)),
              bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.3, nh), target='mu'), silent=True)
        S.communicator = comm
        return S, dict(silent=True), n * n * T * nh, dict(
            workload='HyperStudy 512x512 grid, Laplace observation model, 128 sigma values cint(0, 0.3), T=128, full fit', grid=[n, n], T=T, n_hyper=nh, mode='full')
    if name == 'tiny':        # not a benchmark: the CPU test of this file's launcher / exchange / JSON logic (tests/test_bench_contract.py)
        n, T, nh = 24, 10, 6
        S = bl.HyperStudy(silent=True)
        S.loadData(series(4, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 0.3, nh), target='mean'), silent=True)
        S.communicator = comm
        return S, dict(silent=True), n * n * T * nh, dict(workload='tiny HyperStudy (test of bench.py itself)', grid=[n, n], T=T,
                                                           n_hyper=nh, mode='full')
    if name == 'c3':
        n, T = 1024, 2000
        S = bl.Study(silent=True)
        S.loadData(series(3, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 0.03, target='mean'),
                                            bl.tm.GaussianRandomWalk('s2', 0.008, target='std')), silent=True)
        return S, dict(silent=True), n * n * T, dict(workload='C3 Study 1024x1024 grid, T=2000, GRWxGRW, full fit',
                                                      grid=[n, n], T=T, n_hyper=1, mode='full')
    if name == 'c2':
        n, T = 4096, 10000
        S = bl.Study(silent=True)
        x = series(20260927, T)
        S.loadData(np.stack([x, np.ones(T)], 1), silent=True)
        S.set(bl.om.GaussianMean('mean', bl.cint(-8, 8, n)), bl.tm.GaussianRandomWalk('sigma', 0.02, target='mean'),
              silent=True)
        return S, dict(silent=True), n * T, dict(workload='C2 Study 4096-pt 1-D GaussianMean grid, T=10000, full fit',
                                                  grid=[n], T=T, n_hyper=1, mode='full')
    if name == 'c5':
        n, T = 512, 1000
        tchange = np.arange(3, 1000, 4)[:256]            # BASELINE.json says 256 candidates; the range holds 250
        nh = len(tchange)
        x = series(5, T)
        x[500:] += 2.0
        S = bl.ChangepointStudy(silent=True)
        S.loadData(x, silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.ChangePoint('tChange', tchange), silent=True)
        S.communicator = comm
        return S, dict(silent=True), n * n * T * nh, dict(workload='C5 ChangepointStudy 512x512 grid, T=1000, %d candidate '
                                                           'change-points (arange(3, 1000, 4)), full fit' % nh, grid=[n, n], T=T, n_hyper=nh, mode='full')
    if name in ('c1_hyper', 'coal_hyper1000'):
        # 1-D batches of chains (bl1c::chain1d_kernel).  c1_hyper: BASELINE config C1 as a hyper-study, SURVEY 8(c)'s anchor (coal mining, 200-pt
        # Poisson grid, 20 widths cint(0, 1, 20): logE = -172.6703099789132).  coal_hyper1000: the shape of the reference's hyper-study tutorial
        # (docs/source/tutorials/hyperstudy.ipynb: 1000-pt grid, widths up to 1.0 = 667 grid steps) with 256 widths
        n, nh = (200, 20) if name == 'c1_hyper' else (1000, 256)
        S = bl.HyperStudy(silent=True)
        S.loadExampleData(silent=True)
        S.set(bl.om.Poisson('accident_rate', bl.oint(0, 6, n)),
              bl.tm.GaussianRandomWalk('sigma', bl.cint(0, 1, nh), target='accident_rate'), silent=True)
        S.communicator = comm
        T = len(S.rawData)
        return S, dict(silent=True), n * T * nh, dict(
            workload='HyperStudy coal mining (T=%d), %d-pt Poisson grid x %d random-walk widths cint(0, 1, %d), full fit' % (T, n, nh, nh),
            grid=[n], T=T, n_hyper=nh, mode='full')
    if name == 'coal_breakpoints':
        # the reference's one published heavy workload (docs/source/tutorials/changepointstudy.ipynb, "Analyzing structural breaks":
        # "~25000 individual model fits. It may take several minutes"): coal-mining disasters 1870-1910, a constant rate, a linear
        # decrease with 30 candidate slopes between two break-points at ANY pair of years, a constant rate again -- 23 400 fits
        n = 1000
        S = bl.ChangepointStudy(silent=True)
        S.loadExampleData(silent=True)
        mask = (S.rawTimestamps >= 1870) * (S.rawTimestamps <= 1910)
        S.rawTimestamps = S.rawTimestamps[mask]
        S.rawData = S.rawData[mask]
        S.set(bl.om.Poisson('accident_rate', bl.oint(0, 6, n)),
              bl.tm.SerialTransitionModel(bl.tm.Static(), bl.tm.BreakPoint('t_1', 'all'),
                                          bl.tm.Deterministic(lambda t, slope=np.linspace(-2.0, 0.0, 30): t * slope, target='accident_rate'),
                                          bl.tm.BreakPoint('t_2', 'all'), bl.tm.Static()), silent=True)
        S.communicator = comm
        T, nh = len(S.rawData), 23400
        return S, dict(silent=True), n * T * nh, dict(
            workload='ChangepointStudy coal mining 1870-1910 (reference tutorial changepointstudy.ipynb): 1000-pt Poisson grid, T=41, '
                     'Serial(Static, BreakPoint, Deterministic(30 slopes), BreakPoint, Static), 23400 fits, full fit',
            grid=[n], T=T, n_hyper=nh, mode='full')
    if name == 'fwd2048':
        n, T = 2048, 200
        S = bl.Study(silent=True)
        S.loadData(series(3, T), silent=True)
        S.set(bl.om.Gaussian('mean', bl.cint(-8, 8, n), 'std', bl.oint(0, 4, n)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1', 0.015, target='mean'),
                                            bl.tm.GaussianRandomWalk('s2', 0.004, target='std')), silent=True)
        return S, dict(silent=True, evidenceOnly=True), n * n * T, dict(
            workload='fused forward step, 2048x2048 grid, T=200, evidenceOnly', grid=[n, n], T=T, n_hyper=1,
            mode='evidenceOnly')
    raise ValueError(name)



def measured_traffic(workload, direction):
    """HBM bytes per logical step launch from the committed rocprofv3 PMC passes (separate --pmc runs, tools/prof_workload.sh), newest
    round first; -> (bytes or None, source file or None).  Used when the profiler cannot run inside the bench (--no-pmc, no rocprofv3).
    (Layouts: {workload: {direction: {...}}} since round 3; rounds 1 - 2: 'fwd' / 'bwd' for C4, the workload's name otherwise.)"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for blk in ((d.get(workload) or {}).get(direction), d.get({'forward': 'fwd', 'backward': 'bwd'}[direction]) if workload == 'c4' else None,
                    d.get(workload) if direction == 'forward' else None):
            if isinstance(blk, dict):
                v = blk.get('hbm_bytes_per_step_launch', blk.get('hbm_bytes_per_launch'))
                if v is not None:
                    return v, os.path.relpath(f, ROOT)
    return None, None


def recalibrate(out):
    """The calibrated peak is the best streaming rate OBSERVED in this run: the probe's, or a kernel's real HBM rate when one beats the
    probe (a store-only kernel can) -- then every frac_calibrated of the line is re-expressed against that rate, so none exceeds 1."""
    hbm_blocks = []

    def walk(o):
        if isinstance(o, dict):
            if 'achieved_GBs' in o and 'frac_calibrated' in o:
                hbm_blocks.append(o)
            for v in o.values():
                walk(v)
    walk(out)
    roof = out.get('roofline') or {}
    probe = roof.get('peak_calibrated')
    if not probe or not hbm_blocks:
        return
    best = max([probe] + [b['achieved_GBs'] for b in hbm_blocks])
    if best > probe:
        for b in hbm_blocks:
            b['frac_calibrated'] = b['achieved_GBs'] / best
        roof['peak_calibrated_probe'] = probe
        roof['peak_calibrated'] = best
        roof['peak_calibrated_from'] += '; a kernel of this run streamed faster than the probe: its rate is the calibrated peak'
        if roof.get('frac_calibrated') is not None and roof.get('unit') == 'GB/s':
            roof['frac_calibrated'] = roof['achieved'] / best


def golden_log_evidence(name):
    """logEvidence of the REFERENCE for this exact workload (tests/golden/bench_<name>.npz, generated by importing the reference in
    the build container: tests/golden/gen_bench_golden.py), or None."""
    for cand in ('bench_' + name, 'bench_' + name.replace('_evidence', ''), 'bench_' + name + '_full', name + '_full'):
        f = os.path.join(ROOT, 'tests', 'golden', cand + '.npz')
        if os.path.exists(f):
            return float(np.load(f)['logEvidence'])
    return None


def registered_bound(name):
    """The bound the parity gate holds `log_evidence_rel_err` (the user-visible S.logEvidence) of a workload to: 1e-9, or the bound
    tests/tolerances.py registers for it with its reason (one entry: the published break-point study, COAL_NOISE_CHAINS)."""
    try:
        import tolerances
        return float(getattr(tolerances, 'BENCH_LOG_EVIDENCE_BOUND', {}).get(name, 1e-9))
    except Exception:
        return 1e-9


def rel_err(got, want):
    if want is None:
        return None
    return abs(got - want) / abs(want)


KERNEL_NAMES = {0: 'blk::step_kernel', 1: 'blf::fast_step_kernel', 2: 'blf::fast_step_kernel', 4: 'bl1f::fused1d_kernel (8 time steps per launch)',
                3: 'blm::mfma_step_kernel (+ blf::fast_step_kernel for the radius-0 bucket)',
                5: 'blr::resident_kernel (ONE launch for all time steps; a logical launch = one time step of it)',
                6: 'blc::chain_kernel (rounds of 8 chains resident in LDS for a whole pass; a logical launch = one time step of all '
                   'chains of a batch; the backward kernel of a hyper-study also folds the posteriors into the average posterior)',
                8: 'bl1p::persist1d_kernel (1-D grids: ONE persistent launch per pass; a logical launch = one time step of it)',
                9: 'bl1c::chain1d_kernel (1-D batches: one block per chain runs the whole pass; a logical launch = one time step of all chains)'}


def roofline_of(timing, units, peak_cal=None, pmc=None):
    """Per pass (forward / backward) of the last fit: the three rooflines of its step kernel from the library's HIP-event timing of
    its own stream.  `units` = cells x steps x chains this rank's fit processed; a pass runs all of them in the HIP-event time of
    all its launches, whatever the batch sizes.  cells_per_launch = the average logical launch (one time step of one batch).
    pmc: {'forward': bytes per logical launch, ...} measured with the PMC counters (overrides the designed bytes)."""
    out = {}
    for key, short, stream_bytes in (('forward', 'fwd', BYTES_FWD), ('backward', 'bwd', BYTES_BWD)):
        n = timing.get(key + '_launches', 0)
        ms = timing.get(key + '_ms', 0.0)
        if not (n and ms > 0):
            continue
        per_launch_s = ms * 1e-3 / n
        cells_per_launch = units / n
        rate = cells_per_launch / per_launch_s                      # cell-steps per second of this pass
        variant = timing.get(short + '_kernel_variant', 0)
        designed = timing.get(short + '_hbm_bytes', 0.0) / units
        real, src = designed, 'by construction (blhip_timing.%s_hbm_bytes)' % short
        if pmc and pmc.get(key):
            real, src = pmc[key]['bytes'] / cells_per_launch, pmc[key]['source']
        flop = timing.get(short + '_flops', 0.0) / units
        hbm = dict(bytes_per_cell_step=real, designed_bytes_per_cell_step=designed, source=src, achieved_GBs=real * rate / 1e9,
                   frac_spec=real * rate / 1e9 / HBM_PEAK_GBS,
                   frac_calibrated=(real * rate / 1e9 / peak_cal) if peak_cal else None)
        fp64 = dict(flop_per_cell_step=flop, achieved_TFLOPs=flop * rate / 1e12, peak=FP64_PEAK_TFLOPS,
                    frac=flop * rate / 1e12 / FP64_PEAK_TFLOPS)
        se = dict(bytes_per_cell_step=stream_bytes, GBs_equiv=stream_bytes * rate / 1e9, frac_spec=stream_bytes * rate / 1e9 / HBM_PEAK_GBS)
        out[key] = dict(kernel='%s<%s> (one logical step launch = all launches of one time step of a batch)' % (KERNEL_NAMES.get(variant, 'step_kernel'), key),
                        variant=int(variant), launches=int(n), avg_launch_us=per_launch_s * 1e6, cells_per_launch=cells_per_launch,
                        bound='hbm' if hbm['frac_spec'] >= fp64['frac'] else 'fp64', hbm=hbm, fp64=fp64, streaming_equiv=se)
    return out


def pmc_traffic_in_run(workload, launches_per_dir, timeout=300):
    """HBM bytes per logical step launch of this workload's kernels from the PMC counters, collected NOW: two separate rocprofv3
    passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass; only --kernel-trace beside --pmc) over a one-fit child run of this
    file, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 on gfx950; KiB units).  -> {'forward': {...}, 'backward':
    {...}} or None (no rocprofv3 on PATH, a failed pass, BLHIP_BENCH_PMC=0)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get('BLHIP_BENCH_PMC', '1') == '0' or shutil.which('rocprofv3') is None:
        return None
    tot = {}
    tmp = tempfile.mkdtemp(prefix='blhip_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '-f', 'csv', '-d', d, '-o', 'p', '--', sys.executable,
                   os.path.abspath(__file__), '--workload', workload, '--steps', '1', '--warmup', '0', '--no-extra', '--no-cpu',
                   '--no-pmc', '--no-e2e']
            env = dict(os.environ, TMPDIR='/tmp', BLHIP_BENCH_PMC='0')
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            if r.returncode != 0:
                return None
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row['Counter_Name'] != counter:
                        continue
                    k = kernel_direction(row['Kernel_Name'])
                    if k:
                        tot.setdefault(k, {}).setdefault(counter, 0.0)
                        tot[k][counter] += float(row['Counter_Value'])
        out = {}
        for k, c in tot.items():
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c and launches_per_dir.get(k):
                fb, wb = c['FETCH_SIZE'] * 1024 * 2, c['WRITE_SIZE'] * 1024
                out[k] = dict(bytes=(fb + wb) / launches_per_dir[k], fetch_bytes=fb / launches_per_dir[k], write_bytes=wb / launches_per_dir[k],
                              source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes inside this bench run (FETCH_SIZE x 2, KiB)')
        return out or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_direction(name):
    """'forward' / 'backward' of a step kernel from its demangled name (template arguments), None for the helpers."""
    def targ(tag, i):
        return name.split(tag)[1].split('>')[0].split(',')[i].strip()
    try:
        if 'resident_kernel<' in name:                 # blr::resident_kernel<TR, TC, SEG, CHK, BWD, EVID>
            return 'backward' if targ('resident_kernel<', 4) in ('true', '1') else 'forward'
        if 'chain_fold2_kernel<' in name:              # blc::chain_fold2_kernel<NK, NTW>: backward pass + fused fold, two chains per block
            return 'backward'
        if 'chainax_kernel<' in name:                  # blc::chainax_kernel<NK, NTW, BWD, STORE>: walks on both parameters (blhip_chainax.hpp)
            return 'backward' if targ('chainax_kernel<', 2) in ('true', '1') else 'forward'
        if 'chain_kernel<' in name:                    # blc::chain_kernel<NK, NTW, BWD, STORE>
            return 'backward' if targ('chain_kernel<', 2) in ('true', '1') else 'forward'
        if 'fused1d_kernel<' in name:
            return 'backward' if targ('fused1d_kernel<', -1) in ('true', '1') else 'forward'
        if 'chain1d_kernel<' in name or 'persist1d_kernel<' in name:      # bl1c::chain1d_kernel<OM, BWD, M, CL>, bl1p::persist1d_kernel<OM, BWD>
            tag = 'chain1d_kernel<' if 'chain1d_kernel<' in name else 'persist1d_kernel<'
            return 'backward' if targ(tag, 1) in ('true', '1') else 'forward'
        if 'step_kernel<' in name:                     # <OM, MODE, ...>: MODE 0 = forward
            return 'forward' if targ('step_kernel<', 1) == '0' else 'backward'
    except Exception:
        pass
    return None


def run_workload(bl, name, steps, warmup, comm, barrier):
    S, kw, units, desc = make_study(bl, name, comm)
    for _ in range(warmup):
        S.fit(**kw)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        S.fit(**kw)
    bl.get_engine().synchronize()
    barrier()
    dt = time.perf_counter() - t0
    return S, units, desc, dt


def run_steady(bl, name, min_fits=1, warm_s=0.05, timed_s=0.03, max_fits=60):
    """A bench workload in the steady state: warm-up fits until `warm_s` seconds of them have run (at least `min_fits`), then timed fits
    for at least `timed_s` seconds (at least `min_fits`).  -> (study, units, description, seconds per fit, warm-up fits, timed fits).
    Why: after the idle stretch in which a study is set up the chip needs ~25 ms of load before its kernels run at their steady rate --
    the 2048^2 forward launch (200 steps) takes 1.96, 1.93, 1.91, ... ms and settles at 1.69 ms with the 13th launch
    (profiles/r06d_fwd2048_kernel_trace.csv); three warm-up fits of 2 ms measured the ramp, not the path."""
    S, kw, units, desc = make_study(bl, name, None)
    eng = bl.get_engine()
    n_warm, t0 = 0, time.perf_counter()
    while n_warm < min_fits or (time.perf_counter() - t0 < warm_s and n_warm < max_fits):
        S.fit(**kw)
        n_warm += 1
    eng.synchronize()
    n, t0 = 0, time.perf_counter()
    while n < min_fits or (time.perf_counter() - t0 < timed_s and n < max_fits):
        S.fit(**kw)
        n += 1
    eng.synchronize()
    return S, units, desc, (time.perf_counter() - t0) / n, n_warm, n


def end_to_end(bl, S, kw, units):
    """Fits with everything the reference hands back materialised on the HOST (core.py:356, 408: posteriorSequence is a host
    array there): fit + D2H of the (T, *gridSize) posterior sequence over PCIe into a page-locked numpy array.  Two rounds: the
    first read-back of a size lands in an ordinary (pageable) array while the engine pins a block of that size in the background
    (`cold_ms`); the second one gets that block (`ms`, `value`).  Not part of the
    headline `value`."""
    eng = bl.get_engine()
    rounds = []
    for _ in range(2):
        eng.synchronize()
        t0 = time.perf_counter()
        S.fit(**kw)
        post = None if kw.get('evidenceOnly') else S.posteriorSequence
        rounds.append(time.perf_counter() - t0)
        nbytes = 0 if post is None else int(post.nbytes)
        del post
        S.posteriorSequence = None          # the array goes back to the pool
        if nbytes == 0:
            break
        if hasattr(eng, '_pinned'):
            eng._pinned.wait_ready()        # (the block the first read-back asked for is pinned in the background)
    dt = rounds[-1]
    return dict(ms=dt * 1e3, cold_ms=rounds[0] * 1e3, value=units / dt, unit='grid-cells*timesteps/s', d2h_bytes=nbytes,
                includes='fit() + posteriorSequence copied to the host over PCIe; ms: into a page-locked array of the engine pool (the '
                         'second and later read-backs of a size), cold_ms: the first one, into a pageable array')


def cold_first_fit(workload='c4', timeout=300):
    """Wall time of the FIRST fit of a workload in a fresh process (library load + code-object load of ~1500 kernels + buffer
    allocation + the fit) and of the second one: what a user's first call pays.  -> dict or None."""
    import subprocess
    code = ("import sys, time, json; sys.path.insert(0, %r); import bench; import bayesloop_amd as bl\n"
            "t0 = time.perf_counter(); eng = bl.get_engine(); S, kw, u, d = bench.make_study(bl, %r)\n"
            "S.fit(**kw); eng.synchronize(); t1 = time.perf_counter()\n"
            "S.fit(**kw); eng.synchronize(); t2 = time.perf_counter()\n"
            "print(json.dumps(dict(first_fit_ms=(t1 - t0) * 1e3, second_fit_ms=(t2 - t1) * 1e3)))" % (ROOT, workload))
    try:
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        return json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else dict(error=(r.stderr or '')[-200:])
    except Exception as e:     # noqa: BLE001 -- a diagnostic
        return dict(error=repr(e))


def _cpu_share(args):
    """One worker of the CPU baseline: the oracle's hyper_fit over its share of the sigma values -> seconds of compute."""
    n, T, sigmas = args
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    from oracle import bl_oracle as orc
    g = orc.Grid([orc.cint(-8, 8, n), orc.oint(0, 4, n)])
    data = orc.moving_window(series(4, T), 1)
    prior = orc.compute_prior(g, orc.jeffreys('gaussian'))
    hv, pv, const = orc.hyper_grid([np.asarray(sigmas)], [None])
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        orc.hyper_fit(g, 'gaussian', data, np.arange(T), prior, [('grw', 0)], hv, pv, const)
    return time.perf_counter() - t0


def cpu_baseline(nh=8, T=160, n=512, max_procs=16):
    """The CPU oracle (numpy restatement of the reference path) on a bounded sample of the C4 workload.

    The reference's fit() is single-threaded; its HyperStudy can spread hyper-grid points over processes (nJobs, core.py:
    1317-1326).  Both are timed: one core on `nh` sigma values, then min(host cores, max_procs, free GB) processes with two sigma
    values each (a shorter series keeps a worker under ~0.5 GB; on the 256-core GPU host 16 processes reach 2.4e8, 64
    processes only 1.1e8: memory-bound numpy).  `value` is the better of the two."""
    all_sig = np.linspace(0.0, 0.3, 512)
    dt1 = _cpu_share((n, T, all_sig[::512 // nh][:nh]))
    one = dict(value=n * n * T * nh / dt1, cores=1,
               sample='oracle/bl_oracle.py hyper_fit: %dx%d grid, %d sigma values, T=%d, full fit, %.1f s' % (n, n, nh, T, dt1))
    out = dict(value=one['value'], unit='grid-cells*timesteps/s', cores=1, kind='port', sample=one['sample'], one_core=one)
    procs = min(os.cpu_count() or 1, max_procs)
    try:                                  # ~0.5 GB per worker (T = 96 posteriors of a 512 x 512 grid + accumulators)
        import psutil
        procs = max(1, min(procs, int(psutil.virtual_memory().available / 2 ** 30)))
    except Exception:
        procs = min(procs, 16)
    if procs > 1:
        try:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            Tm, per = 96, 2
            pick = np.linspace(0, 511, per * procs).astype(int)            # sigma values spread over the whole hyper-grid
            shares = [(n, Tm, all_sig[pick[w::procs]].tolist()) for w in range(procs)]
            with ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context('spawn')) as pool:
                times = list(pool.map(_cpu_share, shares, timeout=600))
            val = n * n * Tm * per * procs / max(times)
            multi = dict(value=val, cores=procs,
                         sample='%d processes x %d sigma values, %dx%d grid, T=%d, full fit, slowest worker %.1f s' % (procs, per, n, n, Tm, max(times)))
            out['all_cores'] = multi
            if val > out['value']:
                out.update(value=val, cores=procs, sample=multi['sample'])
        except Exception as e:          # the one-core figure stands
            out['all_cores'] = dict(error=repr(e))
    return out


LINE_LIMIT = 8192          # the driver parses the LAST stdout line; round 4's 35 KB line was not parsed: the line is kept under this


def _sig(x, n=6):
    """Numbers of the compact line at n significant digits (the full-precision record is bench_detail.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not np.isfinite(x):
            return None
        return float('%.*g' % (n, x))
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _kernel_brief(rf):
    """{'fwd_us', 'bwd_us', 'frac_hbm_real', 'frac_streaming_equiv', 'frac_fp64'} of the slower pass (per-pass detail: bench_detail.json)."""
    out = {}
    for key, short in (('forward', 'fwd'), ('backward', 'bwd')):
        if key in (rf or {}):
            out[short + '_us'] = rf[key]['avg_launch_us']
    if rf:
        dom = max(rf.values(), key=lambda r: r['avg_launch_us'] * r['launches'])
        out.update(frac_hbm_real=dom['hbm']['frac_spec'], frac_streaming_equiv=dom['streaming_equiv']['frac_spec'], frac_fp64=dom['fp64']['frac'])
    return out


def compact_line(out):
    """The ONE stdout line: the contract fields, `roofline` and `cpu_baseline` in full meaning but short form, one small record per
    extra workload.  Everything else (per-pass kernel records, configs of the extras, the exchange diagnostic's detail) is in
    bench_detail.json.  Always < LINE_LIMIT characters (asserted; tests/test_bench_contract.py)."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'log_evidence', 'log_evidence_reference', 'log_evidence_rel_err', 'resident_fallbacks', 'device')
    line = {k: out[k] for k in keep if k in out}
    line['device'] = str(line.get('device', ''))[:48]
    r = out.get('roofline')
    if r:
        line['roofline'] = {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_hbm_real', 'frac_fp64',
                                                   'frac_streaming_equiv', 'peak_calibrated', 'frac_calibrated', 'bytes_per_cell_step',
                                                   'avg_launch_us', 'cells_per_launch')}
        kname = str(r.get('kernel', ''))
        line['roofline']['kernel'] = kname.split(' (')[0].replace('<forward>', '').replace('<backward>', '') + ('<backward>' if '<backward>' in kname else '<forward>')
        line['roofline']['traffic_from'] = 'pmc in this run' if 'inside this bench run' in str(r.get('traffic_from')) else (
            'committed pmc pass' if r.get('traffic') is not None else None)
    else:
        line['roofline'] = None
    line['kernels'] = _kernel_brief(out.get('kernels'))
    c = out.get('cpu_baseline')
    if c:
        line['cpu_baseline'] = {k: c.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample')}
        if isinstance(c.get('one_core'), dict):
            line['cpu_baseline']['one_core_value'] = c['one_core'].get('value')
    e = out.get('end_to_end')
    if e:
        line['end_to_end'] = {k: e.get(k) for k in ('ms', 'cold_ms', 'value', 'd2h_bytes', 'error') if e.get(k) is not None}
    if 'exchange' in out:
        x = out['exchange']
        line['exchange'] = {k: (v if not isinstance(v, dict) else {kk: v.get(kk) for kk in ('wall_ms_max_over_ranks', 'device_ms_rank0', 'bytes', 'error')
                                                                   if v.get(kk) is not None}) for k, v in x.items()}
    if 'per_rank' in out:
        line['per_rank'] = {k: v for k, v in out['per_rank'].items() if k != 'note'}
    if 'extra' in out:
        ex = {}
        for name, v in out['extra'].items():
            if 'error' in v:
                ex[name] = dict(error=str(v['error'])[:120])
                continue
            b = dict(value=v['value'], ms_per_step=v['ms_per_step'], log_evidence_rel_err=v.get('log_evidence_rel_err'),
                     resident_fallbacks=v.get('resident_fallbacks'))
            b.update(_kernel_brief(v.get('kernels')))
            for k in ('log_evidence_rel_err_per_chain', 'log_evidence_rel_err_bound', 'speedup_vs_reference_wall', 'host_ms', 'kernel_ms'):
                if v.get(k) is not None:
                    b[k] = v[k]
            if isinstance(v.get('end_to_end'), dict) and 'value' in v['end_to_end']:
                b['end_to_end_value'] = v['end_to_end']['value']
            ex[name] = b
        line['extra'] = ex
    if isinstance(out.get('cold_first_fit'), dict) and 'first_fit_ms' in out['cold_first_fit']:
        line['cold_first_fit_ms'] = out['cold_first_fit']['first_fit_ms']
    line['detail'] = 'bench_detail.json'
    line = _sig(line)
    # full-precision where the driver or the parity gate reads it
    for k in ('value', 'ms_per_step', 'log_evidence', 'log_evidence_reference', 'log_evidence_rel_err'):
        if k in out:
            line[k] = out[k]
    text = json.dumps(line, separators=(',', ':'))
    if len(text) >= LINE_LIMIT:                # never lose the headline: drop the side records, largest first
        for k in ('extra', 'exchange', 'end_to_end', 'kernels'):
            if k in line:
                line[k] = 'see ' + line['detail']
                text = json.dumps(line, separators=(',', ':'))
                if len(text) < LINE_LIMIT:
                    break
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_detail(out):
    """The full record: bench_detail.json beside this file (and under gpurun_out/ when that directory exists, so a gpurun call brings it
    back)."""
    text = json.dumps(out, indent=1)
    where = []
    for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, 'bench_detail.json'), 'w') as f:
                    f.write(text + '\n')
                where.append(os.path.join(d, 'bench_detail.json'))
            except OSError:
                pass
    # (stderr carries one short note only: the driver stores a bounded tail of stdout + stderr, and a 35 KB record there could push
    #  the line out of it)
    try:
        sys.stderr.write('bench.py: full record in %s\n' % (', '.join(os.path.relpath(w, ROOT) for w in where) or 'nowhere (read-only tree)'))
        sys.stderr.flush()
    except Exception:
        pass


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU ordinal), pass rank 0's stdout through.
    A rank that dies takes the others down with it (a survivor would wait in the RCCL bootstrap for ever)."""
    import socket
    import subprocess
    import uuid
    if not TEST_DOUBLE:
        from bayesloop_amd import _abi
        have = _abi.load().blhip_device_count()
        if have < args.gpus:
            sys.exit('bench.py: --gpus %d but only %d HIP device(s) are visible' % (args.gpus, have))
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), BLHIP_RDZV_KEY='bench_' + uuid.uuid4().hex, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    if 'MASTER_PORT' not in env:
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        env['MASTER_PORT'] = str(sk.getsockname()[1])
        sk.close()
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                rc = rc or p.returncode
    for p in live:                       # only reached with a failed rank: stop the rest
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c4')
    ap.add_argument('--no-extra', action='store_true')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='do not run the two rocprofv3 PMC passes for roofline.traffic')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end (PCIe-inclusive) fit')
    ap.add_argument('--opt', action='append', default=[], metavar='KEY=VALUE',
                    help='engine option for an experiment (blhip_set_option), e.g. --opt max_batch=32; recorded in config.engine_options')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    os.environ.setdefault('BLHIP_DEVICE', str(local_rank))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    import bayesloop_amd as bl
    comm = None
    if TEST_DOUBLE:
        # CPU test of THIS file (tests/test_bench_contract.py: launcher environment, self-launch, max-over-ranks timing, the JSON
        # line last on stdout) with the test doubles that live under tests/: the oracle engine + a gloo transport.  Never a
        # measurement: only the `tiny` workload is accepted.
        if args.workload != 'tiny':
            sys.exit('bench.py: BLHIP_BENCH_TEST_DOUBLE=1 runs the workload `tiny` only')
        import bench_double
        eng, comm = bench_double.install(bl, rank, world)
    else:
        eng = bl.get_engine()
        for kv in args.opt:
            k, v = kv.split('=', 1)
            eng.set_option(k, float(v))
        if world > 1 or os.environ.get('BLHIP_FORCE_DIST') == '1':      # (the latter: the RCCL path with a single rank, tests)
            comm = bl.dist.RcclCommunicator(eng, rank=rank, world=world)

    def barrier():
        eng.synchronize()
        if comm is not None:
            comm.barrier()

    for kv in os.environ.get('BLHIP_OPTS', '').split(','):      # tuning experiments: BLHIP_OPTS=key=value,key=value
        if '=' in kv:
            eng.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    peak_cal = eng.bandwidth_probe() if rank == 0 else None
    S, units, desc, dt = run_workload(bl, args.workload, args.steps, args.warmup, comm, barrier)
    if comm is not None:
        dt = comm.allreduce_max(dt)
    timing = dict(S.lastTiming)

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = units * args.steps / dt
        my_units = units / world
        # HBM traffic of the kernels from the PMC counters: collected now (two rocprofv3 passes over a one-fit child run) when the
        # profiler is installed, else the committed passes of the newest round under profiles/
        pmc = None
        if world == 1 and not args.no_pmc:
            nb = max(1, int(timing.get('batches', 1)))
            per_dir = {k: nb * desc['T'] for k in ('forward', 'backward')}
            pmc = pmc_traffic_in_run(args.workload, per_dir)
        if pmc is None and world == 1 and args.workload in ('c4', 'fwd2048'):
            pmc = {}
            for key in ('forward', 'backward'):
                v, src = measured_traffic(args.workload, key)
                if v is not None and (args.workload == 'c4' or key == 'forward'):
                    pmc[key] = dict(bytes=v, source='committed PMC passes: ' + src)
        rf = roofline_of(timing, my_units, peak_cal, pmc)
        dom = max(rf.values(), key=lambda r: r['avg_launch_us'] * r['launches']) if rf else None
        roof = None
        if dom:
            key = 'backward' if 'backward' in dom['kernel'] else 'forward'
            traffic = pmc[key]['bytes'] if pmc and pmc.get(key) else None
            # `achieved` is what the memory system (or the fp64 pipe) REALLY delivers for the dominant kernel: the resident kernels keep
            # the state in LDS, so pricing them at the streaming formulation's bytes (16 / 32 B per cell-step) gives a rate
            # no memory could deliver (it is kept, clearly named, in `algorithmic` -- comparable with the lines of earlier rounds)
            on_hbm = dom['bound'] == 'hbm'
            roof = dict(bound='hbm' if on_hbm else 'mfma',
                        achieved=dom['hbm']['achieved_GBs'] if on_hbm else dom['fp64']['achieved_TFLOPs'],
                        peak=HBM_PEAK_GBS if on_hbm else FP64_PEAK_TFLOPS, unit='GB/s' if on_hbm else 'TFLOP/s',
                        frac=dom['hbm']['frac_spec'] if on_hbm else dom['fp64']['frac'],
                        bytes_per_cell_step=dom['hbm']['bytes_per_cell_step'], bytes_from=dom['hbm']['source'],
                        traffic=traffic, traffic_from=(pmc[key]['source'] if traffic is not None else None),
                        hbm=dom['hbm'], fp64=dom['fp64'],
                        frac_hbm_real=dom['hbm']['frac_spec'], frac_fp64=dom['fp64']['frac'], frac_streaming_equiv=dom['streaming_equiv']['frac_spec'],
                        peak_calibrated=peak_cal, frac_calibrated=dom['hbm']['frac_calibrated'],
                        peak_calibrated_from='blhip_bandwidth_probe: best of a 16-B-per-lane streaming copy (read + write) and a store-only fill of 1 GiB',
                        algorithmic=dict(dom['streaming_equiv'], note='SURVEY 8(d) accounting: the bytes a kernel that streams the state through '
                                         'HBM would move, divided by this kernel\'s time -- an equivalent rate, may exceed the peak'),
                        kernel=dom['kernel'], avg_launch_us=dom['avg_launch_us'], cells_per_launch=dom['cells_per_launch'])
        gold = golden_log_evidence(args.workload)
        out = dict(metric='grid-cells*timesteps/sec (fit())', value=value, unit='grid-cells*timesteps/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
                   scaling='strong', vs_baseline=None, dtype='f64', data='synthetic',
                   config=dict(desc, parallelism='hyper-grid points dealt round-robin to %d GPU(s), one RCCL gather + one reduce' % world,
                               cpu_baseline='sampled (bounded subset of this workload, see cpu_baseline.sample)',
                               **({'engine_options': list(args.opt)} if args.opt else {})),
                   log_evidence=float(S.logEvidence), log_evidence_reference=gold,
                   log_evidence_rel_err=rel_err(float(S.logEvidence), gold), roofline=roof, kernels=rf, device=eng.device_name(),
                   resident_fallbacks=int(timing.get('resident_fallbacks', 0)))
    if rank == 0 and world == 1 and not args.no_e2e:
        kw = dict(silent=True, evidenceOnly=True) if args.workload in ('c4_evidence', 'fwd2048') else dict(silent=True)
        try:
            out['end_to_end'] = end_to_end(bl, S, kw, units)
        except Exception as e:
            out['end_to_end'] = dict(error=repr(e))
    S._posterior_pending = None      # results stay on the device; nothing more is copied back
    eng.release_posterior()
    wedged = False
    if comm is not None and not TEST_DOUBLE and desc.get('mode') == 'full':
        # the accumulator merge on its own, both ways (every rank takes part; zero-filled accumulators of the workload's shape):
        # what the exchange costs next to one rank's share of the chains -- measured only when the driver runs N > 1.  A diagnostic:
        # it runs under a watchdog and whatever happens in it, the line with the timed result is still printed.
        from bayesloop_amd.dist import _bounded
        exch = dict(reduce_ms_in_last_fit=comm.reduce_ms())
        T_, G_ = int(desc['T']), int(np.prod(desc['grid']))

        def one_mode(mode):
            eng.set_option('comm_reduce_mode', mode)
            eng.accum_begin(T_, G_)
            eng.accum_rescale(0.0)
            best = None
            for _ in range(3):
                comm.barrier()
                t0 = time.perf_counter()
                comm.reduce_accumulator(eng, 0)
                wall = comm.allreduce_max(time.perf_counter() - t0) * 1e3
                best = wall if best is None else min(best, wall)
            res = dict(wall_ms_max_over_ranks=best, device_ms_rank0=comm.reduce_ms(), bytes=T_ * G_ * 8)
            eng.accum_end()
            return res
        for mode, label in ((0, 'ncclReduce'), (1, 'reduce_scatter_then_send_to_root')):
            if wedged:
                break
            try:
                exch[label] = _bounded(lambda m=mode: one_mode(m), float(os.environ.get('BLHIP_BENCH_DIAG_TIMEOUT', '60')), 'exchange diagnostic')
            except Exception as e:      # a diagnostic must not lose the line
                exch[label] = dict(error=repr(e))
                wedged = 'did not return' in repr(e)       # a collective hangs: nothing more through this communicator
        if not wedged:
            eng.set_option('comm_reduce_mode', 0)
        if rank == 0:
            out['exchange'] = exch
    if rank == 0 and world > 1:
        out['per_rank'] = dict(note='this rank (0) only: device time of its share', total_ms=timing.get('total_ms'),
                               forward_ms=timing.get('forward_ms'), backward_ms=timing.get('backward_ms'))
    if rank == 0:
        if not args.no_extra and world == 1:
            extra = {}
            for name in ('c4_evidence', 'fwd2048', 'c3', 'c2', 'c5', 'c4_both_axes', 'c4_rows1024', 'c4_wide', 'c4_laplace', 'coal_breakpoints', 'c1_hyper', 'coal_hyper1000'):
                if name == args.workload:
                    continue
                try:
                    # (steady state: see run_steady -- at least 50 ms of warm-up fits, then at least 30 ms of timed ones; three of each for the
                    #  workloads of a few milliseconds; the value is the mean of the timed fits, the kernel times the last one's)
                    few = name in ('fwd2048', 'c1_hyper', 'coal_hyper1000', 'c2', 'coal_breakpoints')
                    import contextlib
                    import io
                    with contextlib.redirect_stdout(io.StringIO()):      # (the study classes print the reference's warnings: stdout carries the line only)
                        S2, u2, d2, dt2, n_warm2, n_timed2 = run_steady(bl, name, 3 if few else 1)
                    tm = dict(S2.lastTiming)
                    g2 = golden_log_evidence(name)
                    extra[name] = dict(value=u2 / dt2, ms_per_step=dt2 * 1e3, log_evidence=float(S2.logEvidence),
                                       log_evidence_reference=g2, log_evidence_rel_err=rel_err(float(S2.logEvidence), g2),
                                       config=d2, kernels=roofline_of(tm, u2, peak_cal),
                                       resident_fallbacks=int(tm.get('resident_fallbacks', 0)))
                    # what of a fit() is kernels (HIP events on the library's stream: forward + backward + fold launches) and what is the rest -- host
                    # code on both sides of the C-ABI, uploads, read-backs of the sums, launch gaps
                    k_ms = float(tm.get('forward_ms', 0.0)) + float(tm.get('backward_ms', 0.0)) + float(tm.get('accumulate_ms', 0.0))
                    extra[name]['kernel_ms'] = k_ms
                    extra[name]['host_ms'] = dt2 * 1e3 - k_ms
                    extra[name]['warmup_fits'] = n_warm2
                    extra[name]['timed_fits'] = n_timed2
                    if name == 'c3':
                        extra[name]['end_to_end'] = end_to_end(bl, S2, dict(silent=True), u2)
                    if name == 'coal_breakpoints':
                        # the reference's one published heavy workload ("may take several minutes"): beside the GPU's fit the wall
                        # time of the REFERENCE ITSELF for the same study, recorded when the fixture was generated in the build
                        # container (tests/golden/gen_bench_golden.py: 8 worker processes), and the study's published number
                        gf = np.load(os.path.join(ROOT, 'tests', 'golden', 'bench_coal_breakpoints_full.npz'))
                        extra[name]['reference'] = dict(seconds=float(gf['reference_seconds']), processes=int(gf['reference_jobs']),
                                                        where='build container (not this box), reference imported from source',
                                                        log10_evidence=float(gf['logEvidence']) / np.log(10),
                                                        log10_evidence_published=-30.63948,
                                                        published_in='docs/source/tutorials/changepointstudy.ipynb')
                        extra[name]['speedup_vs_reference_wall'] = float(gf['reference_seconds']) / dt2
                        # parity per chain (the average model's evidence includes a handful of chains the reference decides by
                        # rounding noise: tests/tolerances.py COAL_NOISE_CHAINS)
                        gl, rl = gf['logEvidenceList'], np.asarray(S2.logEvidenceList, dtype=float)
                        both = np.isfinite(gl) & np.isfinite(rl)
                        # `log_evidence_rel_err` stays the user-visible figure (S.logEvidence of the average model) for every workload;
                        # the per-chain maximum has a name of its own
                        extra[name]['log_evidence_rel_err_per_chain'] = float(np.max(np.abs(rl[both] - gl[both]) / np.abs(gl[both])))
                        extra[name]['log_evidence_rel_err_per_chain_is'] = 'max over the %d chains that run through on both sides' % int(both.sum())
                        extra[name]['log_evidence_rel_err_bound'] = registered_bound(name)
                        extra[name]['chains_stopped'] = dict(reference=int((~np.isfinite(gl)).sum()), here=int((~np.isfinite(rl)).sum()),
                                                             note='registered exception COAL_NOISE_CHAINS')
                    S2._posterior_pending = None
                    eng.release_posterior()
                    del S2
                except Exception as e:     # a failed side workload must not lose the headline line
                    extra[name] = dict(error=repr(e))
            out['extra'] = extra
        if not args.no_extra and world == 1 and not TEST_DOUBLE:
            out['cold_first_fit'] = cold_first_fit(args.workload)
        if not args.no_cpu and world == 1:
            out['cpu_baseline'] = cpu_baseline()
        recalibrate(out)

    def drain_c_stdio():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    if comm is not None and not wedged:
        drain_c_stdio()                          # every rank, before the barrier: nothing of theirs can follow rank 0's line
        comm.barrier()
        comm.close()
    # RCCL prints a version banner through C stdio; on a pipe it sits in libc's buffer until exit, i.e. AFTER anything
    # Python printed.  Drain it first so that the JSON line really is the last line on stdout.
    drain_c_stdio()
    if rank == 0:
        bad = [k for k, v in [(args.workload, out)] + list(out.get('extra', {}).items())
               if isinstance(v, dict) and v.get('log_evidence_rel_err') is not None and v['log_evidence_rel_err'] > registered_bound(k)]
        bad += [k + ' (per chain)' for k, v in out.get('extra', {}).items()
                if isinstance(v, dict) and v.get('log_evidence_rel_err_per_chain') is not None and v['log_evidence_rel_err_per_chain'] > 1e-9]
        write_detail(out)
        sys.stdout.flush()
        print(compact_line(out), flush=True)     # the ONE JSON line, last on stdout, < LINE_LIMIT characters
        if bad:
            sys.exit('bench.py: log-evidence differs from the reference by more than 1e-9 relative: %s' % bad)
    if wedged:                                   # a helper thread sits in a collective that will never return: leave without joining it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
