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

Transports: :class:`RcclCommunicator` (one process per GPU: RCCL through libblhip.so) and :class:`LocalGroup` (ONE process driving
several GPUs with one host thread each -- ``HyperStudy.fit(nJobs=N)``: shared memory + peer copies over xGMI, no RCCL).  The host logic in
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


# ---- rendezvous of the 128-byte RCCL unique id: files in a PRIVATE directory of this user on the node -----------------------
# Protocol (every file is created O_EXCL | O_NOFOLLOW with mode 0600 under a temporary name and renamed into place):
#   rank 0   removes what an earlier job may have left under this key, publishes  id    = [nonce N | unique id]
#   rank r   draws a token R_r of its own, reads id, publishes                    ack.r = [N | R_r]
#   rank 0   waits for every ack that carries ITS nonce, publishes                go    = [N | R_1 ... R_(n-1)]
#   rank r   proceeds only once go carries the nonce it read AND its own token; anything else (a left-over id or go file of a
#            crashed job with the same key, read before rank 0 replaced it) is ignored and polled past.
# A stale or planted file therefore never reaches ncclCommInitRank; the communicator init itself is bounded by a timeout.
NONCE_BYTES = 16
ID_BYTES = 128


def _rendezvous_dir():
    """A directory only this user can write: $BLHIP_RDZV_DIR, else $XDG_RUNTIME_DIR/blhip, else /tmp/blhip-<uid> (0700, owner and
    mode verified, symlinks refused)."""
    d = os.environ.get('BLHIP_RDZV_DIR')
    if d is None:
        base = os.environ.get('XDG_RUNTIME_DIR')
        d = os.path.join(base, 'blhip') if base and os.path.isdir(base) and os.access(base, os.W_OK) else '/tmp/blhip-%d' % os.getuid()
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise BackendError('rendezvous directory %s is not a private directory of this user (owner %d, mode %o)'
                           % (d, st.st_uid, st.st_mode & 0o777))
    return d


def _rendezvous_key(key=None):
    """Every rank of one job derives the same key: an explicit one (``BLHIP_RDZV_KEY``; bench.py's self-launch sets a fresh one per
    run); under a launcher that exports MASTER_ADDR / MASTER_PORT (torch.distributed.run and friends) those + the restart count
    + the run id; else the parent's pid (ranks forked by one launcher process)."""
    if key is None:
        key = os.environ.get('BLHIP_RDZV_KEY')
    if key is None and os.environ.get('MASTER_PORT'):
        key = 'm%s_%s_%s_%s' % (os.environ.get('MASTER_ADDR', 'localhost'), os.environ['MASTER_PORT'],
                                os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'))
    if key is None:
        key = 'p%d' % os.getppid()
    return ''.join(c if c.isalnum() or c in '-_.' else '_' for c in str(key))


def _rendezvous_path(key=None):
    global _comm_counter
    _comm_counter += 1              # (a per-process counter keeps several communicators of one job apart)
    return os.path.join(_rendezvous_dir(), 'rdzv_%s_%d' % (_rendezvous_key(key), _comm_counter))


def _publish(path, payload):
    tmp = '%s.tmp%d' % (path, os.getpid())
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
    try:
        os.write(fd, payload)
    finally:
        os.close(fd)
    os.replace(tmp, path)


def _read(path, nbytes):
    try:
        fd = os.open(path, os.O_RDONLY | getattr(os, 'O_NOFOLLOW', 0))
    except OSError:
        return None
    try:
        raw = os.read(fd, nbytes + 1)
    finally:
        os.close(fd)
    return raw if len(raw) == nbytes else None


def exchange_unique_id(lib, rank, world, key=None, timeout=300.0):
    """Rank 0 creates the id (``blhip_comm_unique_id`` = ncclGetUniqueId); -> (id bytes, list of files rank 0 removes once every
    rank has joined the communicator)."""
    if world == 1:
        buf = C.create_string_buffer(ID_BYTES)
        if lib.blhip_comm_unique_id(buf) != 0:
            raise BackendError(lib.blhip_last_error(None).decode())
        return buf.raw, []
    path = _rendezvous_path(key)
    ack = lambda r: '%s.ack%d' % (path, r)
    go = path + '.go'
    t_start = time.time()

    def expired(what):
        if time.time() - t_start > timeout:
            raise BackendError('rank %d: %s at %s after %.0f s (are all %d ranks running with the same rendezvous key?)'
                               % (rank, what, path, timeout, world))
        time.sleep(0.005)

    if rank == 0:
        for f in [path, go] + [ack(r) for r in range(1, world)]:          # left-overs of an earlier job with this key
            try:
                os.unlink(f)
            except OSError:
                pass
        buf = C.create_string_buffer(ID_BYTES)
        if lib.blhip_comm_unique_id(buf) != 0:
            raise BackendError(lib.blhip_last_error(None).decode())
        nonce = os.urandom(NONCE_BYTES)
        _publish(path, nonce + buf.raw)
        tokens = {}
        while len(tokens) < world - 1:
            for r in range(1, world):
                if r not in tokens:
                    raw = _read(ack(r), 2 * NONCE_BYTES)
                    if raw is not None and raw[:NONCE_BYTES] == nonce:
                        tokens[r] = raw[NONCE_BYTES:]
            if len(tokens) < world - 1:
                expired('only %d of %d ranks answered' % (len(tokens), world - 1))
        _publish(go, nonce + b''.join(tokens[r] for r in range(1, world)))
        return buf.raw, [path, go] + [ack(r) for r in range(1, world)]
    token = os.urandom(NONCE_BYTES)
    acked = None
    while True:
        raw = _read(path, NONCE_BYTES + ID_BYTES)
        if raw is not None:
            nonce = raw[:NONCE_BYTES]
            if nonce != acked:
                _publish(ack(rank), nonce + token)
                acked = nonce
            g = _read(go, NONCE_BYTES * world)
            if g is not None and g[:NONCE_BYTES] == nonce and g[NONCE_BYTES * rank:NONCE_BYTES * (rank + 1)] == token:
                return raw[NONCE_BYTES:], []
        expired('no confirmed RCCL unique id')


def _bounded(fn, timeout, what):
    """Run fn() in a helper thread and give up after `timeout` seconds (ncclCommInitRank waits for ever for a rank that never
    comes); the stuck thread is left behind -- the caller is expected to end the process."""
    import threading
    box = {}

    def run():
        try:
            box['value'] = fn()
        except BaseException as e:          # noqa: BLE001 -- handed to the caller
            box['error'] = e
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        raise BackendError('%s did not return within %.0f s' % (what, timeout))
    if 'error' in box:
        raise box['error']
    return box.get('value')


class RcclCommunicator:
    """RCCL communicator of this process's GPU context, one rank per GPU (ranks / world size from the arguments or from the
    launcher's RANK / WORLD_SIZE).  Collective: every rank of the job must construct it."""

    def __init__(self, engine=None, rank=None, world=None, key=None, uid=None):
        from . import engine as _engine_mod
        self.engine = engine if engine is not None else _engine_mod.get_engine()
        if not hasattr(self.engine, 'lib') or not hasattr(self.engine, 'ctx'):
            raise BackendError('RcclCommunicator needs the HIP engine (one libblhip context per GPU)')
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.size = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        lib = self.engine.lib
        uid, published = exchange_unique_id(lib, self.rank, self.size, key) if uid is None else (uid, [])
        init_timeout = float(os.environ.get('BLHIP_COMM_INIT_TIMEOUT', '300'))
        _bounded(lambda: self.engine._check(lib.blhip_comm_init(self.engine.ctx, uid, self.size, self.rank)), init_timeout,
                 'rank %d: ncclCommInitRank (world %d)' % (self.rank, self.size))
        self._open = True
        _bounded(self.barrier, init_timeout, 'rank %d: the first collective' % self.rank)     # everyone has joined: the rendezvous files can go
        for f in published:
            try:
                os.remove(f)
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

    def reduce_ms(self):
        """HIP-event time (ms) of this rank's last reduce_accumulator."""
        ms = C.c_double()
        self.engine._check(self.engine.lib.blhip_comm_timing(self.engine.ctx, C.byref(ms)))
        return ms.value

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
        # (LocalGroup with two contexts on ONE physical device -- a test configuration: their fits take turns)
        turn = comm.device_lock(engine) if getattr(getattr(comm, 'group', None), 'shared', False) else _nullcontext()
        with turn:
            # (one rank: its share is the whole hyper-grid -- no gather of a 16-MB op-value matrix through an index array)
            ops = np.asarray(op_values)
            res = engine.fit(problem, ops if size == 1 else ops[mine], forward_only=forward_only, evidence_only=evidence_only,
                             keep_posterior=False, accumulate=want_post, log_chain_weight=log_w if size == 1 else log_w[mine], owner=owner)
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
            if rank == 0 and size > 1 and gstats is not None:
                # checksum of the merge: the per-step sums of the merged accumulator must be the sum of the per-step sums every rank
                # reported BEFORE the exchange (they travelled with the gather).  A peer copy / reduce that dropped, duplicated or
                # mis-ordered a slice cannot pass; costs one pass over the accumulator on the root.
                merged = engine.accum_row_stats(problem)
                want = np.asarray(gstats)
                # column 0 (sum A > 0) relatively; the mean columns (sum A grid_k) against the scale sum A x max|grid_k|: on a grid
                # symmetric about 0 they cancel to rounding noise, which two summation orders do not reproduce relatively
                gmax = np.array([1.0] + [max(float(np.max(np.abs(m))), 1e-300) for m in problem.marginal])
                bad = ~(np.abs(merged - want) <= 1e-9 * np.abs(want[:, :1]) * gmax[None, :] + 1e-300)
                if np.any(bad):
                    from .engine import BackendError
                    t_bad = int(np.argmax(np.any(bad, axis=1)))
                    raise BackendError('accumulator merge over %d devices failed its checksum (first bad time step %d: merged %r, '
                                       'sum of the parts %r)' % (size, t_bad, merged[t_bad].tolist(), want[t_bad].tolist()))
        if ok and rank == 0:
            means = engine.accum_finalize(problem)
            from .engine import DevicePosterior
            posterior = DevicePosterior(engine, 1, T, grid_size)     # lazy D2H / device-side reductions of the average
        elif ok:
            means = (gstats[:, 1:] / gstats[:, :1]).T                # core.py:1416-1419 from the gathered per-step sums

    return dict(log_evidence=np.asarray(logE), local_evidence=np.asarray(local), abort_step=np.asarray(astep),
                posterior_mean=means, posterior=posterior, timing=timing)


# ---- several GPUs driven by ONE process: HyperStudy.fit(nJobs = N) ------------------------------------------------------------
class LocalGroup:
    """The in-process stand-in for the reference's process pool (bayesloop/core.py:1317-1326): N engines (one libblhip context per
    device), N host threads (ctypes releases the GIL inside blhip_fit, so the N devices compute at the same time), and the same
    ONE gather + ONE accumulator merge as the multi-process path -- through shared memory and peer copies over xGMI
    (blhip_accum_peer_reduce / _gather) instead of RCCL.  ``member(r)`` is the communicator object of thread r."""

    def __init__(self, engines):
        import threading
        self.engines = list(engines)
        self.size = len(self.engines)
        self.barrier = threading.Barrier(self.size)
        self.slots = [None] * self.size
        # two contexts on ONE physical device (a test configuration: BLHIP_NJOBS_DEVICES=0,0) must not compute at the same time --
        # the resident kernels need every CU of the chip
        self.locks = {}
        for e in self.engines:
            self.locks.setdefault(getattr(e, 'device', id(e)), threading.Lock())
        self.shared = len(self.locks) < self.size

    def member(self, rank):
        return _LocalMember(self, rank)


class _LocalMember:
    def __init__(self, group, rank):
        self.group, self.rank, self.size = group, rank, group.size
        self.collectives = []

    def _wait(self):
        self.group.barrier.wait(timeout=float(os.environ.get('BLHIP_NJOBS_TIMEOUT', '3600')))

    def device_lock(self, engine):
        return self.group.locks[getattr(engine, 'device', id(engine))]

    def all_gather(self, a):
        g = self.group
        g.slots[self.rank] = np.array(a, dtype=np.float64, copy=True)
        self._wait()
        out = [np.array(s, copy=True) for s in g.slots]
        self._wait()                      # (nobody overwrites a slot before everyone has read it)
        self.collectives.append(('all_gather', int(np.size(a))))
        return out

    def reduce_accumulator(self, engine, root=0):
        """Reduce-scatter over time slices, then the root fetches the slices: every device takes in (N - 1) / N of ONE accumulator
        over its N - 1 links (xGMI is point-to-point) instead of the root taking in N - 1 whole ones."""
        g = self.group
        engine.synchronize()
        self._wait()                      # every accumulator is rescaled to the common exponent and idle
        T = engine.accum_shape()[0]
        bounds = [(int(b[0]), int(b[-1]) + 1) if len(b) else (0, 0) for b in np.array_split(np.arange(T), self.size)]
        others = [g.engines[j] for j in range(self.size) if j != self.rank]
        with self.device_lock(engine) if g.shared else _nullcontext():
            engine.accum_peer_reduce(others, *bounds[self.rank])
        self._wait()                      # every slice is complete on its owner
        if self.rank == root:
            srcs = [j for j in range(self.size) if j != root]
            engine.accum_peer_gather([g.engines[j] for j in srcs], [bounds[j] for j in srcs])
        self._wait()                      # (the owners' buffers were read: they may be reused now)
        self.collectives.append(('reduce', int(T)))

    def barrier(self):
        self._wait()


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_MULTI_GPU_OFF = []          # reasons; non-empty: the in-process multi-GPU path is off for the rest of the process


def multi_gpu_strict():
    return os.environ.get('BLHIP_NJOBS_MULTI_GPU', '').lower() == 'strict'


def multi_gpu_disabled():
    return bool(_MULTI_GPU_OFF) or os.environ.get('BLHIP_NJOBS_MULTI_GPU', '1').lower() in ('0', 'off', 'no')


def disable_multi_gpu(reason):
    """The in-process multi-GPU path failed: say so ONCE on stderr, keep fitting on the root device."""
    import sys
    if not _MULTI_GPU_OFF:
        sys.stderr.write('[bayesloop_amd] fit(nJobs > 1) over several GPUs failed (%s); this fit is repeated on one GPU and the '
                         'process keeps using one GPU (BLHIP_NJOBS_MULTI_GPU=strict to raise instead)\n' % reason)
    _MULTI_GPU_OFF.append(str(reason))


def local_devices(n_jobs, root_device=0):
    """Devices of an in-process sharded fit: ``BLHIP_NJOBS_DEVICES`` (comma-separated ordinals, duplicates allowed: a test
    configuration) or the first min(n_jobs, visible) devices starting with the root engine's."""
    env = os.environ.get('BLHIP_NJOBS_DEVICES')
    if env:
        return [int(x) for x in env.split(',') if x.strip() != ''][:max(1, n_jobs)]
    from . import _abi
    have = _abi.load().blhip_device_count()
    devs = [root_device] + [d for d in range(have) if d != root_device]
    return devs[:max(1, min(n_jobs, have))]


def local_sharded_hyper_fit(engines, problem, op_values, prior_values, forward_only=False, evidence_only=False, owner=None):
    """sharded_hyper_fit on len(engines) devices from ONE process: engines[0] is the root (its context ends up holding the
    finalised average posterior), one host thread per further engine.  Returns the root's result dict (+ 'per_rank_timing')."""
    import threading
    group = LocalGroup(engines)
    results, errors = [None] * group.size, []

    def work(r):
        comm = group.member(r)
        try:
            results[r] = sharded_hyper_fit(engines[r], problem, op_values, prior_values, comm, forward_only=forward_only,
                                           evidence_only=evidence_only, owner=owner if r == 0 else None)
        except BaseException as e:        # noqa: BLE001 -- re-raised in the calling thread
            errors.append((r, e))
            group.barrier.abort()         # the others must not wait for this thread for ever
    threads = [threading.Thread(target=work, args=(r,), name='blhip-njobs-%d' % r) for r in range(1, group.size)]
    for t in threads:
        t.start()
    work(0)
    for t in threads:
        t.join()
    for r in range(1, group.size):        # the other devices' accumulators are spent -- also when a thread failed (up to 16 GiB each)
        try:
            engines[r].accum_end()
        except Exception:                 # noqa: BLE001
            pass
    if errors:
        import threading as _t
        first = [e for r, e in sorted(errors, key=lambda x: x[0]) if not isinstance(e, _t.BrokenBarrierError)]
        raise (first[0] if first else errors[0][1])
    out = results[0]
    out['per_rank_timing'] = [dict(res['timing']) if res else {} for res in results]
    for r, tm in enumerate(out['per_rank_timing']):      # how each device fetched the others' accumulator slices (set by the merge, after its fit)
        try:
            tm['peer_copy_path'] = engines[r].last_timing().get('peer_copy_path', 0)
        except Exception:                 # noqa: BLE001
            pass
    return out
