"""
Multi-GPU HyperStudy: one process per GPU, hyper-grid points dealt out round-robin (rank r runs points r, r + R, ...),
no communication while the chains run, and ONE exchange at the end over RCCL / xGMI (``torch.distributed`` backend
"nccl" is RCCL on ROCm).  The reference's ``HyperStudy._parallelFit`` hands out the contiguous chunks of
``np.array_split`` (bayesloop/core.py:1464-1465); the results do not depend on which worker runs which point, but the
cost of a chain does depend on its hyper-parameter value (a wider random walk is a wider stencil): on the C4 workload the
last contiguous chunk takes 16 % longer than the first one, the strided shares are within 0.5 % of each other
(tools/shard_balance.py).

  1. all-gather of the packed per-chain scalars  [logEvidence | localEvidence (T) | abort step]      (KBs)
  2. only when posteriors were requested: all-reduce(MAX) of one scalar (the accumulators' reference exponents),
     a local rescale, and a reduce(SUM) of the (T, G) float64 accumulator to rank 0 -- the linear-space equivalent of
     the reference's ``np.logaddexp`` merge of the sub-studies (core.py:1335-1340) followed by ``-= amax; exp``
     (core.py:1375-1376).

The same code path runs with the "gloo" backend on CPU tensors (tests, world_size 2).
"""
from __future__ import annotations

import numpy as np


class TorchCommunicator:
    """torch.distributed process group as the transport (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        backend = dist.get_backend(group)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
        self.device = device

    def new_buffer(self, n):
        return self.torch.empty(int(n), dtype=self.torch.float64, device=self.device)

    def _to_dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def all_gather(self, a):
        """a: float64 array, same shape on every rank -> list of arrays by rank."""
        t = self._to_dev(a)
        out = [self.torch.empty_like(t) for _ in range(self.size)]
        self.dist.all_gather(out, t, group=self.group)
        return [o.cpu().numpy() for o in out]

    def allreduce_max(self, x):
        t = self._to_dev(np.array([x], dtype=np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.cpu().numpy()[0])

    def reduce_sum(self, buf, dst=0):
        """In-place SUM-reduce of a device buffer (torch tensor) to rank ``dst``."""
        if self.device.type == 'cuda':
            self.torch.cuda.synchronize(self.device)
        self.dist.reduce(buf, dst=dst, op=self.dist.ReduceOp.SUM, group=self.group)
        if self.device.type == 'cuda':
            self.torch.cuda.synchronize(self.device)

    def broadcast(self, a, src=0):
        t = self._to_dev(a)
        self.dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def barrier(self):
        self.dist.barrier(group=self.group)


def default_communicator():
    """A TorchCommunicator if torch.distributed is initialised with more than one rank, else None."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return TorchCommunicator()
    return None


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
    posterior (callable returning the (T, *gridSize) average posterior, or None on non-root ranks / evidence-only),
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

    buf = None
    if want_post:
        buf = comm.new_buffer(T * G) if comm is not None else None
        engine.accum_begin(T, G, external=buf, owner=owner)

    n_mine = len(mine)
    logE = np.zeros(n_mine)
    local = np.zeros((n_mine, T))
    astep = np.full(n_mine, -1.0)
    timing = {}
    if n_mine > 0:
        res = engine.fit(problem, np.asarray(op_values)[mine], forward_only=forward_only, evidence_only=evidence_only,
                         keep_posterior=False, accumulate=want_post, log_chain_weight=log_w[mine], owner=owner)
        logE, local, astep, timing = res.log_evidence, res.local_evidence, res.abort_step.astype(float), res.timing

    if comm is not None:
        # ---- the single gather: [logE | local (T) | abort] per chain, padded to the largest share ----------------
        width = (n_h + size - 1) // size
        packed = np.zeros((width, T + 2))
        packed[:n_mine, 0] = logE
        packed[:n_mine, 1:T + 1] = local
        packed[:n_mine, T + 1] = astep
        parts = comm.all_gather(packed)
        full = np.zeros((n_h, T + 2))
        for r in range(size):
            idx = shard_indices(n_h, size, r)
            full[idx] = parts[r][:len(idx)]
        logE, local, astep = full[:, 0], full[:, 1:T + 1], full[:, T + 1]

    means, posterior = None, None
    if want_post:
        ref, _ = engine.accum_log_ref()
        if comm is not None:
            gref = comm.allreduce_max(ref)
            if np.isfinite(gref):
                engine.accum_rescale(gref)
                comm.reduce_sum(buf, dst=0)
            ok = np.isfinite(gref)
        else:
            ok = np.isfinite(ref)
        if ok and rank == 0:
            means = engine.accum_finalize(problem)
            from .engine import DevicePosterior
            posterior = DevicePosterior(engine, 1, T, grid_size)     # lazy D2H / device-side reductions of the average
        if comm is not None:
            m = comm.broadcast(means if means is not None else np.full((ndim, T), np.nan), src=0)
            means = m if ok else None

    return dict(log_evidence=np.asarray(logE), local_evidence=np.asarray(local), abort_step=np.asarray(astep),
                posterior_mean=means, posterior=posterior, timing=timing)
