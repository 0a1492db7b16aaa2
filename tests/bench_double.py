"""
TEST DOUBLES for bench.py itself (tests/test_bench_contract.py): with BLHIP_BENCH_TEST_DOUBLE=1 bench.py runs its `tiny` workload
on the oracle engine over a gloo transport, so that its launcher handling (RANK / LOCAL_RANK / WORLD_SIZE of the driver's
``python -m torch.distributed.run ... bench.py --gpus N``), its self-launch, the barrier + max-over-ranks timing and the
JSON-line-last logic are executed on a machine without GPUs.  Nothing here is a measurement; the product never imports it.
"""
import os

import numpy as np

from gloo_comm import GlooCommunicator
from oracle_engine import OracleEngine


class BenchOracleEngine(OracleEngine):
    device = 0

    def device_name(self):
        return 'oracle test double (CPU)'

    def set_option(self, key, value):
        pass

    def bandwidth_probe(self, nbytes=0, iterations=0):
        return 1.0

    def last_timing(self):
        return dict(forward_ms=1.0, backward_ms=1.0, forward_launches=10, backward_launches=10, batches=1, total_ms=2.0)

    def fit(self, *a, **k):
        res = super().fit(*a, **k)
        res.timing = self.last_timing()
        return res


class BenchGloo(GlooCommunicator):
    def allreduce_max(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t[0])

    def close(self):
        self.dist.destroy_process_group()


def install(bl, rank, world):
    import torch.distributed as dist
    eng = BenchOracleEngine()
    bl.set_engine(eng)
    comm = None
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://%s:%s' % (os.environ.get('MASTER_ADDR', '127.0.0.1'), os.environ['MASTER_PORT']),
                                rank=rank, world_size=world)
        comm = BenchGloo()
    np.seterr(all='ignore')
    return eng, comm
