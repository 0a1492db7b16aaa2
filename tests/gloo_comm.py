"""
TEST TRANSPORT: the communicator interface of bayesloop_amd.dist (rank, size, all_gather, reduce_accumulator) over
torch.distributed's gloo backend on host arrays, so that the multi-rank host logic of ``sharded_hyper_fit`` (shares, the
single gather with its trailer, the accumulator merge) runs with world_size > 1 on a machine without GPUs.  It pairs with the
oracle engine (tests/oracle_engine.py), whose accumulator is a numpy array.  The product's transport is
``bayesloop_amd.dist.RcclCommunicator`` (RCCL through the C-ABI); nothing under bayesloop_amd/ imports torch.
"""
import numpy as np


class GlooCommunicator:
    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.collectives = []                 # what the exchange consisted of (asserted by the tests)

    def all_gather(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        out = [self.torch.empty_like(t) for _ in range(self.size)]
        self.dist.all_gather(out, t, group=self.group)
        self.collectives.append(('all_gather', t.numel()))
        return [o.numpy() for o in out]

    def reduce_accumulator(self, engine, root=0):
        buf = self.torch.from_numpy(engine.acc_lin)         # shares memory with the test double's linear accumulator
        self.dist.reduce(buf, dst=root, op=self.dist.ReduceOp.SUM, group=self.group)
        self.collectives.append(('reduce', buf.numel()))

    def barrier(self):
        self.dist.barrier(group=self.group)
