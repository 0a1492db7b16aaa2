"""
The compute engine behind ``Study.fit`` / ``HyperStudy.fit``: a problem description in plain arrays and the
MI355X engine that runs it through the C-ABI of libblhip.so.

``FitProblem`` is the drop-in boundary seen from Python: everything ``Study.fit`` (reference bayesloop/core.py:330-486)
reads from the study object, flattened into arrays.  ``HipEngine`` is the only engine the product ships; it raises
:class:`BackendError` when the HIP library or the GPU is missing (no CPU fallback).  Tests may install another object
with the same methods through :func:`set_engine` to exercise the host-side logic on a machine without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _abi
from .exceptions import BackendError


@dataclass
class FitProblem:
    obs_model: int                      # _abi.OM_*
    marginal: List[np.ndarray]          # marginal grid per parameter (core.py:156)
    lattice: List[float]                # lattice constants (core.py:161-166)
    data: np.ndarray                    # formatted data (T, seg[, d]) (core.py:349)
    timestamps: np.ndarray              # formatted timestamps (T,) (core.py:350)
    prior: np.ndarray                   # alpha_0 on the grid (core.py:363)
    ops: List[Tuple]                    # transition program [(kind, axis[, segment, flags])], list order
    reset_prior: Optional[np.ndarray] = None    # what a ChangePoint resets to (transitionModels.py:300-312)
    indep_prior: Optional[np.ndarray] = None    # what Independent restarts from (transitionModels.py:351-360)
    lik: Optional[np.ndarray] = None    # (T, G) host-evaluated likelihood for OM_TABLE
    seg_len: int = 1
    resume_time: float = -1.0           # BLHIP_RESUME: time stamp the transition into step 0 is evaluated at
    carry_slot: int = 0                 # which carried state of the context (OnlineStudy: one per transition model)
    backward_init: Optional[np.ndarray] = None  # the backward message entering the last step (None: uniform, core.py:424-425)

    @property
    def grid_size(self):
        return [len(m) for m in self.marginal]

    @property
    def T(self):
        return len(self.data)

    @property
    def G(self):
        return int(np.prod(self.grid_size))


@dataclass
class FitResult:
    log_evidence: np.ndarray            # (n_chains,)
    local_evidence: np.ndarray          # (n_chains, T)
    posterior_mean: Optional[np.ndarray]    # (n_chains, ndim, T) or None (evidenceOnly)
    abort_step: np.ndarray              # (n_chains,) -1 or step index
    abort_phase: np.ndarray             # (n_chains,) 0 forward / 1 backward
    timing: dict = field(default_factory=dict)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DevicePosterior:
    """Handle of a posterior sequence that still lives in HBM (source 0: kept posterior of a fit, 1: finalised average
    posterior).  Calling it copies the (T, *gridSize) array to the host; ``marginal`` / ``time_average`` reduce on the
    device and copy only the result (the consumers of posteriorSequence in the reference reduce it immediately:
    bayesloop/core.py:886, 915, 979-980)."""

    def __init__(self, engine, source, T, grid_size, chain=0):
        self.engine, self.source, self.T, self.grid_size, self.chain = engine, source, T, list(grid_size), chain

    _hinted = [False]

    def __call__(self):
        # The whole (T, *gridSize) array crosses PCIe here (BASELINE C3: 16 GiB, 14 x the time of the fit itself).  What the reference's
        # users do with posteriorSequence next is almost always a reduction the device has already (getParameterDistributions / plot:
        # marginals; getParameterMeanValues; simulate: time average; getParameterDistribution(t, ...): one row) -- say so once.
        nbytes = 8 * self.T * int(np.prod(self.grid_size))
        if nbytes >= (1 << 30) and not DevicePosterior._hinted[0] and os.environ.get('BLHIP_QUIET', '') != '1':
            DevicePosterior._hinted[0] = True
            import sys
            sys.stderr.write('[bayesloop_amd] copying a %.1f-GiB posteriorSequence from the GPU to the host.  getParameterDistributions / '
                             'getParameterDistribution(t, ...) / getParameterMeanValues / simulate / plot reduce it ON the device and copy only '
                             'their result -- touch S.posteriorSequence itself only if the whole array is needed (BLHIP_QUIET=1 silences this).\n'
                             % (nbytes / 2.0 ** 30))
        if self.source == 0:
            return self.engine.posterior(self.chain, self.T, self.grid_size)
        return self.engine.accum_read(self.T, self.grid_size)

    def row(self, index):
        """One time step of the sequence as an array of gridSize (copies G doubles, not T * G)."""
        if self.source == 0:
            return self.engine.posterior(self.chain, self.T, self.grid_size, t0=index, t1=index + 1)[0]
        return self.engine.accum_read(self.T, self.grid_size, t0=index, t1=index + 1)[0]

    def marginal(self, k):
        if len(self.grid_size) > 2:
            # grids with 3 and more parameters: the device-side reductions are 2-D; reduce on the host, a few time steps at a time
            out = np.empty((self.T, self.grid_size[k]))
            axes = tuple(a + 1 for a in range(len(self.grid_size)) if a != k)
            step = max(1, int(2 ** 26 // max(1, int(np.prod(self.grid_size)))))
            for t0 in range(0, self.T, step):
                t1 = min(self.T, t0 + step)
                rows = (self.engine.posterior(self.chain, self.T, self.grid_size, t0=t0, t1=t1) if self.source == 0
                        else self.engine.accum_read(self.T, self.grid_size, t0=t0, t1=t1))
                out[t0:t1] = rows.sum(axis=axes)
            return out
        return self.engine.marginal(self.source, self.chain, k, self.T, self.grid_size[k])

    def time_average(self):
        return self.engine.time_average(self.source, self.chain, self.grid_size)


class _PinnedPool:
    """Result arrays of the big read-backs (posteriorSequence: (T, *gridSize) float64, 16 GiB for BASELINE C3) in PAGE-LOCKED host
    memory: the D2H copy is then one DMA per 256 MiB piece at the PCIe rate (57 GB/s measured) instead of the runtime's staging
    through pageable memory (25 GB/s).  Pinning itself is slow (16 GiB: 1.1 s, more than the pageable copy), so the FIRST big
    read-back of a size goes to an ordinary array, and a block of that size is pinned in the background once that copy is DONE
    (`copied()`; side by side the two fight over the page tables: the first 16-GiB read-back took 2.2 s instead of 0.7 s); later
    ones get the block.  The arrays are ordinary writable numpy arrays; when the last view of one dies its block comes back here and ONE free
    block (the largest) is kept."""
    MIN_BYTES = 32 << 20

    def __init__(self, lib):
        import threading
        self.lib = lib
        self.free = None                 # (ptr, nbytes)
        self.enabled = os.environ.get('BLHIP_PINNED_RESULTS', '1') != '0'
        self.lock = threading.Lock()
        self.pending = None              # the background thread pinning a block
        self.deferred = 0                # bytes of the block to pin once the pageable read-back in flight is done
        self.released = False            # the engine is gone: blocks that come back are freed, not parked

    def _give_back(self, ptr, nbytes):
        try:
            with self.lock:
                if self.released:
                    old = (ptr, nbytes)
                elif self.free is None or self.free[1] < nbytes:
                    old, self.free = self.free, (ptr, nbytes)
                else:
                    old = (ptr, nbytes)
            if old is not None:
                self.lib.blhip_host_free(old[0])
        except Exception:                # interpreter shutdown
            pass

    def _pin_in_background(self, nbytes):
        import threading
        if self.pending is not None and self.pending.is_alive():
            return

        def work():
            ptr = self.lib.blhip_host_alloc(nbytes)
            if ptr:
                self._give_back(ptr, nbytes)
        self.pending = threading.Thread(target=work, name='blhip-pin', daemon=True)
        self.pending.start()

    def wait_ready(self, timeout=30.0):
        """Block until the background pinning (if any) has finished (bench.py: between its cold and its warm round)."""
        if self.pending is not None:
            self.pending.join(timeout)

    def empty(self, shape):
        shape = [int(x) for x in shape]
        nbytes = int(np.prod(shape)) * 8
        if not self.enabled or nbytes < self.MIN_BYTES:
            return np.empty(shape)
        ptr = None
        with self.lock:
            # (any free block that is large enough: pinning a better-fitting one in the background would cost up to ~1 s of
            #  hipHostRegister only to be freed again when it comes back -- the larger block is the one that is kept)
            if self.free is not None and nbytes <= self.free[1]:
                (ptr, cap), self.free = self.free, None
        if not ptr:
            self.deferred = nbytes
            return np.empty(shape)       # this time through pageable memory
        buf = (C.c_char * cap).from_address(ptr)
        weakref.finalize(buf, self._give_back, ptr, cap)
        return np.frombuffer(buf, dtype=np.float64, count=nbytes // 8).reshape(shape)

    def copied(self):
        """The read-back into the array `empty` handed out last is complete: pin a block for the next one of that size."""
        nbytes, self.deferred = self.deferred, 0
        if nbytes:
            self._pin_in_background(nbytes)

    def release(self):
        self.wait_ready(5.0)
        with self.lock:
            blk, self.free = self.free, None
            self.released = True
        if blk is not None:
            self.lib.blhip_host_free(blk[0])


class HipEngine:
    """One libblhip context on one GPU."""

    name = 'hip'

    def __init__(self, device: int = 0):
        self.lib = _abi.load()
        if self.lib.blhip_device_count() <= 0:
            raise BackendError('no HIP device visible: bayesloop_amd needs an AMD MI355X (gfx950); there is no CPU fallback')
        self.device = device
        self.ctx = self.lib.blhip_create(device)
        if not self.ctx:
            raise BackendError('blhip_create(%d) failed: %s' % (device, self.lib.blhip_last_error(None).decode()))
        self._posterior_owner = None
        self._keep = []
        self._pinned = _PinnedPool(self.lib)
        # engine options from the environment, e.g. BLHIP_ENGINE_OPTS=resident_timeout_s=2 for processes that SHARE a GPU (the
        # resident kernels wait for peer blocks that another process's kernels may keep off the chip for a while)
        for kv in os.environ.get('BLHIP_ENGINE_OPTS', '').split(','):
            if '=' in kv:
                self.set_option(kv.split('=')[0].strip(), float(kv.split('=')[1]))

    def __del__(self):
        try:
            if getattr(self, '_pinned', None) is not None:
                self._pinned.release()
            if getattr(self, 'ctx', None):
                self.lib.blhip_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise BackendError(self.lib.blhip_last_error(self.ctx).decode())

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.blhip_device_name(self.ctx, buf, 256))
        return buf.value.decode()

    def set_option(self, key, value):
        self._check(self.lib.blhip_set_option(self.ctx, key.encode(), float(value)))

    _token_counter = [0]

    def _prior_token(self, a):
        """blhip_problem.prior_token: a name for the CONTENT of a prior array, so that the library can skip uploading it again (32 MiB on
        a 2048 x 2048 grid: 0.6 ms of every fit).  Only arrays that cannot change get one: float64, C-contiguous, owning their
        memory and READ-ONLY -- what Study._computePrior caches per study.  Everything else: 0 (uploaded every time)."""
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.base is None and not a.flags.writeable):
            return 0
        tokens = self.__dict__.setdefault('_prior_tokens', {})
        hit = tokens.get(id(a))
        if hit is not None and hit[0]() is a:
            return hit[1]
        self._token_counter[0] += 1
        tok = self._token_counter[0]
        key = id(a)
        tokens[key] = (weakref.ref(a, lambda _r, k=key, t=tokens: t.pop(k, None)), tok)
        return tok

    def _problem(self, p: FitProblem):
        """-> (ctypes Problem, list of arrays that must stay alive)"""
        ndim = len(p.marginal)
        if not 1 <= ndim <= _abi.MAX_DIM:
            raise BackendError('the MI355X engine supports 1 to %d observation-model parameters (got %d)' % (_abi.MAX_DIM, ndim))
        keep = []
        cp = _abi.Problem()
        cp.ndim = ndim
        cp.obs_model = p.obs_model
        for k in range(ndim):
            m = _f64(p.marginal[k])
            keep.append(m)
            cp.n[k] = len(m)
            cp.marginal[k] = _abi.dptr(m)
            cp.lattice[k] = float(p.lattice[k])
        data = _f64(p.data)
        T = data.shape[0]
        if data.ndim == 1:
            data = data.reshape(T, 1, 1)
        elif data.ndim == 2:
            data = data.reshape(T, data.shape[1], 1)
        elif data.ndim != 3:
            raise BackendError('formatted data must have 1 to 3 dimensions')
        data = np.ascontiguousarray(data)
        ts = _f64(p.timestamps)
        prior = _f64(p.prior).ravel()
        G = int(np.prod([len(m) for m in p.marginal]))
        if prior.size != G or ts.size != T:
            raise BackendError('prior / timestamps do not match grid / data')
        keep += [data, ts, prior]
        cp.prior_token = self._prior_token(p.prior)
        cp.T = T
        cp.seg_len = data.shape[1]
        cp.data_dim = data.shape[2]
        cp.data = _abi.dptr(data)
        cp.timestamps = _abi.dptr(ts)
        cp.prior = _abi.dptr(prior)
        if p.reset_prior is not None:
            rp = _f64(p.reset_prior).ravel()
            keep.append(rp)
            cp.reset_prior = _abi.dptr(rp)
        if p.indep_prior is not None:
            ip = _f64(p.indep_prior).ravel()
            keep.append(ip)
            cp.indep_prior = _abi.dptr(ip)
        if p.lik is not None:
            lik = _f64(p.lik).reshape(T, G)
            keep.append(lik)
            cp.lik = _abi.dptr(lik)
        ops = (_abi.Op * max(1, len(p.ops)))()
        for k, op in enumerate(p.ops):
            ops[k].kind = op[0]
            ops[k].axis = op[1]
            ops[k].segment = op[2] if len(op) > 2 else -1
            ops[k].flags = op[3] if len(op) > 3 else 0
        keep.append(ops)
        cp.n_ops = len(p.ops)
        cp.ops = ops
        cp.resume_time = float(p.resume_time)
        cp.carry_slot = int(p.carry_slot)
        if p.backward_init is not None:
            bi = _f64(p.backward_init).ravel()
            if bi.size != G:
                raise BackendError('backward_init does not match the grid')
            keep.append(bi)
            cp.backward_init = _abi.dptr(bi)
        return cp, keep

    def fit(self, problem: FitProblem, op_values, forward_only=False, evidence_only=False, keep_posterior=False,
            accumulate=False, log_chain_weight=None, owner=None, resume=False, carry=False) -> FitResult:
        """n_chains = len(op_values) independent passes (Study.fit per hyper-grid point)."""
        prev = self._posterior_owner() if self._posterior_owner is not None else None
        if prev is not None and prev is not owner:
            self._posterior_owner = None
            prev._materialize_posterior()      # the previous study's posterior still lives in this context
        cp, keep = self._problem(problem)
        ov = _f64(op_values).reshape(-1, max(1, len(problem.ops))) if len(problem.ops) else np.zeros((len(op_values), 1))
        n = ov.shape[0] if len(problem.ops) else len(op_values)
        T, ndim = problem.T, len(problem.marginal)
        logE = np.zeros(n)
        local = np.zeros((n, T))
        # (per-chain posterior means are not part of a hyper-study's results -- core.py:1416-1419 takes them from the average
        #  posterior -- and the backward kernels skip their sums when nobody asks)
        means = None if (evidence_only or accumulate) else np.zeros((n, ndim, T))
        astep = np.full(n, -1, dtype=np.int64)
        aphase = np.zeros(n, dtype=np.int32)
        res = _abi.Result()
        res.log_evidence = _abi.dptr(logE)
        res.local_evidence = _abi.dptr(local)
        res.posterior_mean = _abi.dptr(means)
        res.abort_step = astep.ctypes.data_as(C.POINTER(C.c_int64))
        res.abort_phase = aphase.ctypes.data_as(C.POINTER(C.c_int32))
        flags = (_abi.FORWARD_ONLY if forward_only else 0) | (_abi.EVIDENCE_ONLY if evidence_only else 0) | \
                (_abi.KEEP_POSTERIOR if keep_posterior else 0) | (_abi.ACCUMULATE if accumulate else 0) | \
                (_abi.RESUME if resume else 0) | (_abi.CARRY if carry else 0)
        lw = None if log_chain_weight is None else _f64(log_chain_weight)
        self._check(self.lib.blhip_fit(self.ctx, C.byref(cp), n, _abi.dptr(ov), _abi.dptr(lw), flags, C.byref(res)))
        del keep
        if keep_posterior and not evidence_only:
            self._posterior_owner = None if owner is None else weakref.ref(owner)
        return FitResult(logE, local, means, astep, aphase, self.last_timing())

    # ---- carried states of streaming fits (OnlineStudy.step) -------------------------------------------------------
    def carry_mix(self, slot, weights, accumulate=False):
        """mix = (accumulate ? mix : 0) + sum_j weights[j] * carried state j of `slot` (stays on the device)."""
        w = _f64(weights)
        self._check(self.lib.blhip_carry_mix(self.ctx, int(slot), len(w), _abi.dptr(w), 1 if accumulate else 0))

    def carry_read(self, slot, chain, grid_size):
        """One chain's carried distribution (chain >= 0) or the mix buffer (chain = -1) as an array of `grid_size`."""
        out = np.empty(list(grid_size))
        self._check(self.lib.blhip_carry_read(self.ctx, int(slot), int(chain), _abi.dptr(out)))
        return out

    def carry_write(self, slot, states):
        """Restore the carried distributions of `slot`: states is (n_chains, *grid_size), rows normalised."""
        st = _f64(states)
        n = st.shape[0]
        st = st.reshape(n, -1)
        self._check(self.lib.blhip_carry_write(self.ctx, int(slot), n, st.shape[1], _abi.dptr(st)))

    def carry_release(self, slot=-1):
        self._check(self.lib.blhip_carry_release(self.ctx, int(slot)))

    def bandwidth_probe(self, nbytes=1 << 30, iterations=20):
        """GB/s (read + write) of a streaming copy on this GPU: the calibrated roofline beside the 8 TB/s spec peak."""
        out = C.c_double()
        self._check(self.lib.blhip_bandwidth_probe(self.ctx, int(nbytes), int(iterations), C.byref(out)))
        return out.value

    def last_timing(self):
        t = _abi.Timing()
        self._check(self.lib.blhip_last_timing(self.ctx, C.byref(t)))
        return t.as_dict()

    def posterior(self, chain, T, grid_size, t0=0, t1=None):
        """Normalised posterior sequence of one chain of the last fit(keep_posterior=True) as (T, *grid_size)
        (or the rows t0 .. t1-1 of it)."""
        t1 = T if t1 is None else t1
        out = self._pinned.empty([t1 - t0] + list(grid_size))
        try:
            self._check(self.lib.blhip_posterior_read(self.ctx, chain, t0, t1, _abi.dptr(out)))
        finally:
            self._pinned.copied()
        return out

    def marginal(self, source, chain, keep_axis, T, n_keep):
        """(T, n_keep) marginal probabilities of one parameter, reduced on the device (source 0: kept posterior of
        `chain`, 1: finalised average posterior)."""
        out = np.empty((T, n_keep))
        self._check(self.lib.blhip_posterior_marginal(self.ctx, source, chain, keep_axis, _abi.dptr(out)))
        return out

    def time_average(self, source, chain, grid_size):
        out = np.empty(list(grid_size))
        self._check(self.lib.blhip_posterior_time_average(self.ctx, source, chain, _abi.dptr(out)))
        return out

    def release_posterior(self, owner=None):
        """Forget who owns the device-resident results (they will not be copied to the host when overwritten)."""
        cur = self._posterior_owner() if self._posterior_owner is not None else None
        if owner is None or cur is owner:
            self._posterior_owner = None
        cur = self._accum_owner() if getattr(self, '_accum_owner', None) is not None else None
        if owner is None or cur is owner:
            self._accum_owner = None

    # ---- average posterior of a hyper-study ----------------------------------------------------------------------
    def accum_begin(self, T, G, owner=None):
        """Starts the evidence-weighted average of a hyper-study in a library-owned (T, G) float64 buffer in HBM."""
        ref = getattr(self, '_accum_owner', None)
        prev = ref() if ref is not None else None
        if prev is not None and prev is not owner:
            self._accum_owner = None
            prev._materialize_posterior()       # the previous study's average posterior still lives in the accumulator
        self._accum_owner = None if owner is None else weakref.ref(owner)
        self._check(self.lib.blhip_accum_begin(self.ctx, T, G, None))
        self._acc_shape = (int(T), int(G))

    def accum_row_stats(self, problem: FitProblem):
        """(T, 1 + ndim) per-step sums [sum A, sum A grid_k] of the not yet finalised accumulator (relative to its
        reference exponent): what a rank contributes to the merged normalisers / posterior means."""
        cp, keep = self._problem(problem)
        out = np.zeros((problem.T, 1 + len(problem.marginal)))
        self._check(self.lib.blhip_accum_row_stats(self.ctx, C.byref(cp), _abi.dptr(out)))
        return out

    def accum_log_ref(self):
        ref = C.c_double()
        n = C.c_int64()
        self._check(self.lib.blhip_accum_state(self.ctx, C.byref(ref), None, C.byref(n)))
        return ref.value, n.value

    def accum_rescale(self, new_log_ref):
        self._check(self.lib.blhip_accum_rescale(self.ctx, float(new_log_ref)))

    def accum_set_owner(self, owner):
        """Whose results the accumulator holds (brought to the host before another study's fit reuses it)."""
        self._accum_owner = None if owner is None else weakref.ref(owner)

    def accum_fold_host(self, posterior, log_weight):
        """Folds one chain whose (T, *gridSize) posterior sequence lives on the host (a hyper-grid point fitted through the transition-model
        plug-in interface) into the open accumulator, weight exp(log_weight) (reference core.py:1358-1366)."""
        post = _f64(posterior)
        T, G = self._acc_shape
        if post.size != T * G:
            raise BackendError('accum_fold_host: the sequence has %d values, the accumulator %d x %d' % (post.size, T, G))
        self._check(self.lib.blhip_accum_fold_host(self.ctx, _abi.dptr(post), float(log_weight)))

    def accum_finalize(self, problem: FitProblem):
        cp, keep = self._problem(problem)
        ndim = len(problem.marginal)
        means = np.zeros((ndim, problem.T))
        self._check(self.lib.blhip_accum_finalize(self.ctx, C.byref(cp), _abi.dptr(means)))
        return means

    def accum_read(self, T, grid_size, t0=0, t1=None):
        t1 = T if t1 is None else t1
        out = self._pinned.empty([t1 - t0] + list(grid_size))
        try:
            self._check(self.lib.blhip_accum_read(self.ctx, t0, t1, _abi.dptr(out)))
        finally:
            self._pinned.copied()
        return out

    def accum_end(self):
        self._check(self.lib.blhip_accum_end(self.ctx))

    def accum_shape(self):
        return self._acc_shape

    # ---- merge of the accumulators of several contexts of THIS process (dist.LocalGroup) -------------------------------------
    def accum_peer_reduce(self, others, row0, row1):
        """acc[row0:row1] += sum of the other engines' acc[row0:row1] (peer copies over xGMI, summed in list order)."""
        arr = (C.c_void_p * max(1, len(others)))(*[o.ctx for o in others])
        self._check(self.lib.blhip_accum_peer_reduce(self.ctx, arr, len(others), int(row0), int(row1)))

    def accum_peer_gather(self, others, bounds):
        """acc[a:b] = other.acc[a:b] for every (other, (a, b))."""
        n = len(others)
        arr = (C.c_void_p * max(1, n))(*[o.ctx for o in others])
        r0 = (C.c_int64 * max(1, n))(*[int(b[0]) for b in bounds])
        r1 = (C.c_int64 * max(1, n))(*[int(b[1]) for b in bounds])
        self._check(self.lib.blhip_accum_peer_gather(self.ctx, arr, n, r0, r1))

    def synchronize(self):
        self._check(self.lib.blhip_synchronize(self.ctx))


_engine = None
_device_engines = {}


def engine_for_device(device):
    """One engine (libblhip context) per device ordinal and process: the process-wide engine for its own device, further ones
    created on first use (HyperStudy.fit(nJobs=N) drives one per GPU).  A list entry repeated in BLHIP_NJOBS_DEVICES (a test
    configuration: two contexts on one GPU) gets a context of its own."""
    root = get_engine()
    if device == getattr(root, 'device', None):
        return root
    if device not in _device_engines:
        _device_engines[device] = HipEngine(device)
    return _device_engines[device]


def extra_engine(device):
    """A further context on `device`, not shared with anybody (tests: several contexts on one GPU)."""
    return HipEngine(device)


def get_engine():
    """The process-wide engine (created on first use on the device chosen by ``BLHIP_DEVICE`` / ``LOCAL_RANK``)."""
    global _engine
    if _engine is None:
        import os
        dev = int(os.environ.get('BLHIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        _engine = HipEngine(dev)
    return _engine


def set_engine(engine):
    """Install an engine object (tests use this to run host-side logic without a GPU).  Returns the previous one."""
    global _engine
    prev, _engine = _engine, engine
    return prev
