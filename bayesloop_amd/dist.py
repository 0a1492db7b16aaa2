"""
Multi-GPU HyperStudy: one process per GPU, hyper-grid points dealt out round-robin (rank r runs points r, r + R, ...),
no communication while the chains run, and ONE exchange at the end over RCCL / xGMI, bound directly through the C-ABI
(``blhip_comm_*`` in include/blhip.h; no PyTorch, no MPI).

What it replaces in the reference: ``HyperStudy.fit(nJobs > 1)`` fans the hyper-grid out with
``pool.map(self._parallelFit, ...)`` (bayesloop/core.py:1317-1326) and merges the sub-studies on the host
(core.py:1335-1340).  ``_parallelFit`` hands out the contiguous chunks of ``np.array_split`` (core.py:1464-1465); the
results do not depend on which worker runs which point, but the cost of a chain does depend on its hyper-parameter
value (a wider random walk is a wider stencil): on the C4 workload the last contiguous chunk takes 16 % longer than the
first one, the strided shares are within 0.5 % of each other (tools/shard_balance.py).

The exchange:
  1. ONE all-gather of the packed per-chain rows [logEvidence | localEvidence (T) | abort step] padded to the largest
     share, plus one trailer per rank: the reference exponent of its average-posterior accumulator and the per-step
     sums of that accumulator ([sum A, sum A grid_k], T x (1 + ndim) doubles).  The maximum of the exponents and the
     posterior means of the merged average follow from the gathered trailers on every rank, so neither a second
     collective for the maximum nor a broadcast of the means is needed.  An evidence-only fit is exactly this gather.
  2. only when posteriors were requested: a local rescale to the common exponent and ONE reduce(SUM) of the (T, G) float64
     accumulator to rank 0, in place in HBM -- the linear-space equivalent of the reference's ``np.logaddexp`` merge
     (core.py:1339) followed by ``-= amax; exp`` (core.py:1375-1376).

Transports: :class:`RcclCommunicator` (the product: RCCL through libblhip.so).  The host logic in
:func:`sharded_hyper_fit` only needs ``rank``, ``size``, ``all_gather(array)`` and ``reduce_accumulator(engine, root)``;
the CPU tests drive it with a gloo-backed stand-in that lives in tests/ (tests/gloo_comm.py).
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from .exceptions import BackendError

SUM, MAX, MIN = 0, 1, 2
_comm_counter = 0


# ---- rendezvous of the 128-byte RCCL unique id: a file on the node (the ranks of one node share /tmp) ---------------------
def _rendezvous_path(key=None):
    """Every rank of one job derives the same path: an explicit key (``BLHIP_RDZV_KEY``; bench.py's self-launch sets a fresh
    one per run) or, under a generic one-process-per-GPU launcher, the launcher's pid (the parent of all
    ranks) + MASTER_PORT + restart count; a per-process counter keeps several communicators of one job apart."""
    global _comm_counter
    _comm_counter += 1
    if key is None:
        key = os.environ.get('BLHIP_RDZV_KEY')
    if key is None:
        key = 'p%d_%s_%s' % (os.getppid(), os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'))
    d = os.environ.get('BLHIP_RDZV_DIR', '/tmp')
    return os.path.join(d, 'blhip_rdzv_%s_%d' % (key, _comm_counter))


def exchange_unique_id(lib, rank, world, key=None, timeout=300.0):
    """Rank 0 creates the id (``blhip_comm_unique_id`` = ncclGetUniqueId) and publishes it atomically; the others poll."""
    nbytes = 128
    if world == 1:
        buf = C.create_string_buffer(nbytes)
        if lib.blhip_comm_unique_id(buf) != 0:
            raise BackendError(lib.blhip_last_error(None).decode())
        return buf.raw, None
    path = _rendezvous_path(key)
    t_start = time.time()
    if rank == 0:
        buf = C.create_string_buffer(nbytes)
        if lib.blhip_comm_unique_id(buf) != 0:
            raise BackendError(lib.blhip_last_error(None).decode())
        tmp = '%s.tmp%d' % (path, os.getpid())
        with open(tmp, 'wb') as f:
            f.write(buf.raw)
        os.replace(tmp, path)
        return buf.raw, path
    while True:
        try:
            # a left-over of an earlier job with the same key cannot be newer than this process minus the launch skew
            if os.path.getmtime(path) >= t_start - 600.0:
                with open(path, 'rb') as f:
                    raw = f.read()
                if len(raw) == nbytes:
                    return raw, None
        except OSError:
            pass
        if time.time() - t_start > timeout:
            raise BackendError('rank %d: no RCCL unique id at %s after %.0f s (is rank 0 running?)' % (rank, path, timeout))
        time.sleep(0.01)


class RcclCommunicator:
    """RCCL communicator of this process's GPU context, one rank per GPU (ranks / world size from the arguments or from the
    launcher's RANK / WORLD_SIZE).  Collective: every rank of the job must construct it."""

    def __init__(self, engine=None, rank=None, world=None, key=None):
        from . import engine as _engine_mod
        self.engine = engine if engine is not None else _engine_mod.get_engine()
        if not hasattr(self.engine, 'lib') or not hasattr(self.engine, 'ctx'):
            raise BackendError('RcclCommunicator needs the HIP engine (one libblhip context per GPU)')
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.size = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        lib = self.engine.lib
        uid, published = exchange_unique_id(lib, self.rank, self.size, key)
        self.engine._check(lib.blhip_comm_init(self.engine.ctx, uid, self.size, self.rank))
        self._open = True
        self.barrier()                     # everyone has joined: the rendezvous file can go
        if published:
            try:
                os.remove(published)
            except OSError:
                pass

    # ---- what sharded_hyper_fit needs ---------------------------------------------------------------------------------
    def all_gather(self, a):
        """a: float64 array, same shape on every rank -> list of arrays by rank (one ncclAllGather)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty((self.size,) + a.shape)
        lib = self.engine.lib
        self.engine._check(lib.blhip_comm_allgather(self.engine.ctx, a.ctypes.data_as(C.POINTER(C.c_double)), a.size,
                                                     out.ctypes.data_as(C.POINTER(C.c_double))))
        return [out[r] for r in range(self.size)]

    def reduce_accumulator(self, engine, root=0):
        """In-place SUM-reduce of the engine's (T, G) average-posterior accumulator to `root` (one ncclReduce in HBM)."""
        if engine is not self.engine:
            raise BackendError('the accumulator lives in another context than this communicator')
        self.engine._check(self.engine.lib.blhip_comm_reduce_accum(self.engine.ctx, int(root)))

    # ---- small host-side collectives (bench.py: barrier, max-over-ranks timing) ------------------------------------------
    def allreduce(self, values, op=SUM):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self.engine._check(self.engine.lib.blhip_comm_allreduce(self.engine.ctx, v.ctypes.data_as(C.POINTER(C.c_double)),
                                                                 v.size, int(op)))
        return v

    def allreduce_max(self, x):
        return float(self.allreduce([x], MAX)[0])

    def barrier(self):
        self.engine.synchronize()
        self.allreduce([0.0], SUM)         # returns after every rank's contribution has arrived (stream-synchronised)

    def info(self):
        w, r, v = C.c_int(), C.c_int(), C.c_int()
        self.engine._check(self.engine.lib.blhip_comm_info(self.engine.ctx, C.byref(w), C.byref(r), C.byref(v)))
        return dict(world=w.value, rank=r.value, rccl_version=v.value)

    def close(self):
        if getattr(self, '_open', False):
            self._open = False
            self.engine._check(self.engine.lib.blhip_comm_destroy(self.engine.ctx))


def shard_indices(n, size, rank):
    """Hyper-grid points of one rank: rank, rank + size, ... (balances any smooth cost trend over the hyper-grid)."""
    return np.arange(rank, n, size)


def chunk_bounds(n, size):
    """Contiguous near-equal chunks, identical to np.array_split(range(n), size)."""
    parts = np.array_split(np.arange(n), size)
    return [(int(p[0]), int(p[-1]) + 1) if len(p) else (0, 0) for p in parts]


def sharded_hyper_fit(engine, problem, op_values, prior_values, comm, forward_only=False, evidence_only=False,
                      owner=None):
    """
    Runs the chains of a hyper-study on this rank's share of the hyper-grid and merges the results.
    Returns a dict: log_evidence (n_h,), local_evidence (n_h, T), abort_step (n_h,), posterior_mean (ndim, T) or None,
    posterior (handle of the (T, *gridSize) average posterior, or None on non-root ranks / evidence-only),
    timing (dict of this rank's last device timing).
    """
    n_h = len(op_values)
    T, G = problem.T, problem.G
    grid_size = list(problem.grid_size)
    ndim = len(grid_size)
    size = 1 if comm is None else comm.size
    rank = 0 if comm is None else comm.rank
    mine = shard_indices(n_h, size, rank)
    with np.errstate(divide='ignore'):
        log_w = np.log(np.asarray(prior_values, dtype=float))
    want_post = not evidence_only

    if want_post:
        engine.accum_begin(T, G, owner=owner)

    n_mine = len(mine)
    logE = np.zeros(n_mine)
    local = np.zeros((n_mine, T))
    astep = np.full(n_mine, -1.0)
    timing = {}
    if n_mine > 0:
        res = engine.fit(problem, np.asarray(op_values)[mine], forward_only=forward_only, evidence_only=evidence_only,
                         keep_posterior=False, accumulate=want_post, log_chain_weight=log_w[mine], owner=owner)
        logE, local, astep, timing = res.log_evidence, res.local_evidence, res.abort_step.astype(float), res.timing

    ref = -np.inf
    if want_post:
        ref, _ = engine.accum_log_ref()
    gref, gstats = ref, None
    if comm is not None:
        # ---- the single gather: [logE | local (T) | abort] per chain, padded to the largest share, + the trailer ----------
        width = (n_h + size - 1) // size
        ntrail = (1 + T * (1 + ndim)) if want_post else 0
        packed = np.zeros(width * (T + 2) + ntrail)
        rows = packed[:width * (T + 2)].reshape(width, T + 2)
        rows[:n_mine, 0] = logE
        rows[:n_mine, 1:T + 1] = local
        rows[:n_mine, T + 1] = astep
        if want_post:
            packed[width * (T + 2)] = ref
            if np.isfinite(ref):
                packed[width * (T + 2) + 1:] = engine.accum_row_stats(problem).ravel()
        parts = comm.all_gather(packed)
        full = np.zeros((n_h, T + 2))
        for r in range(size):
            idx = shard_indices(n_h, size, r)
            full[idx] = parts[r][:width * (T + 2)].reshape(width, T + 2)[:len(idx)]
        logE, local, astep = full[:, 0], full[:, 1:T + 1], full[:, T + 1]
        if want_post:
            refs = np.array([parts[r][width * (T + 2)] for r in range(size)])
            gref = float(np.max(refs))
            if np.isfinite(gref):
                with np.errstate(over='ignore', invalid='ignore'):
                    scale = np.where(np.isfinite(refs), np.exp(refs - gref), 0.0)
                gstats = sum(scale[r] * parts[r][width * (T + 2) + 1:].reshape(T, 1 + ndim) for r in range(size)
                             if scale[r] > 0.0)

    means, posterior = None, None
    if want_post:
        ok = bool(np.isfinite(gref))
        if ok and comm is not None:
            engine.accum_rescale(gref)
            comm.reduce_accumulator(engine, root=0)
        if ok and rank == 0:
            means = engine.accum_finalize(problem)
            from .engine import DevicePosterior
            posterior = DevicePosterior(engine, 1, T, grid_size)     # lazy D2H / device-side reductions of the average
        elif ok:
            means = (gstats[:, 1:] / gstats[:, :1]).T                # core.py:1416-1419 from the gathered per-step sums

    return dict(log_evidence=np.asarray(logE), local_evidence=np.asarray(local), abort_step=np.asarray(astep),
                posterior_mean=means, posterior=posterior, timing=timing)
