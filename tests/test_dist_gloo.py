"""The multi-rank HyperStudy path (bayesloop_amd.dist.sharded_hyper_fit) with world_size 2 and 3 on CPU: round-robin shares, the
single gather (rows + accumulator trailer) and the accumulator merge must reproduce the unsharded golden result.  Transport:
tests/gloo_comm.py (gloo, host arrays) + the oracle engine; the product transport (RCCL through the C-ABI) runs in the -m gpu tests."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(%(root)r, 'tests')); sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=%(world)d)
    import bayesloop_amd as bl
    import cases, compare, oracle_adapter as oa
    from oracle_engine import OracleEngine
    from gloo_comm import GlooCommunicator
    bl.set_engine(OracleEngine())
    for case in %(cases)r:
        S = cases.build(bl, case)
        S.communicator = GlooCommunicator()
        with np.errstate(all='ignore'):
            S.fit(**cases.fit_kwargs(case))
        gold = oa.load_golden(case)
        res = dict(logEvidence=S.logEvidence, localEvidence=S.localEvidence, logEvidenceList=np.array(S.logEvidenceList),
                   hyperParameterDistribution=S.hyperParameterDistribution)
        if not cases.CASES[case].get('fit', {}).get('evidenceOnly'):
            res['posteriorMeanValues'] = S.posteriorMeanValues
            if dist.get_rank() == 0:
                res['posteriorSequence'] = S.posteriorSequence
            else:
                assert S.posteriorSequence is None
                gold = {k: v for k, v in gold.items() if not k.startswith('posteriorS') and not k.startswith('posteriorR')
                        and not k.startswith('marginalS')}
        compare.check(res, gold, dict(compare.ORACLE_TOL, post_rtol=1e-10, small_rtol=1e-10))
        kinds = [k for k, _ in S.communicator.collectives]
        evid = bool(cases.CASES[case].get('fit', {}).get('evidenceOnly'))
        assert kinds == (['all_gather'] if evid else ['all_gather', 'reduce']), kinds      # ONE gather (+ ONE reduce)
        n = bl.get_engine().fits
        print('rank', dist.get_rank(), case, 'ok; chains fitted on this rank so far:', n)
    dist.barrier()
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_hyperstudy_matches_unsharded(tmp_path, world):
    pytest.importorskip('torch')
    script = tmp_path / 'worker.py'
    names = ['kat_hyper_1hp', 'c4_2hp', 'c4_small_evidence', 'c5_cp_grw', 'c1_coal_hyper']
    script.write_text(WORKER % dict(root=ROOT, port=free_port(), world=world, cases=names))
    env = dict(os.environ, OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\\n%s' % (r, out)
        assert out.count(' ok;') == len(names), out


def test_chunks_equal_array_split():
    from bayesloop_amd.dist import chunk_bounds
    for n in (1, 2, 7, 16, 512, 513):
        for size in (1, 2, 3, 4, 8):
            parts = np.array_split(np.arange(n), size)
            got = chunk_bounds(n, size)
            for p, (a, b) in zip(parts, got):
                assert list(p) == list(range(a, b))


def test_round_robin_shares_partition_the_hyper_grid():
    from bayesloop_amd.dist import shard_indices
    for n in (1, 2, 7, 16, 512, 513):
        for size in (1, 2, 3, 4, 8):
            shares = [shard_indices(n, size, r) for r in range(size)]
            assert sorted(np.concatenate(shares).tolist()) == list(range(n))
            assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
            assert max(len(s) for s in shares) == (n + size - 1) // size


RANDOM_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(%(root)r, 'tests')); sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=%(world)d)
    import bayesloop_amd as bl
    import cases, compare, oracle_adapter as oa, random_cases
    from oracle_engine import OracleEngine
    from gloo_comm import GlooCommunicator
    bl.set_engine(OracleEngine())
    done = 0
    for seed in range(%(seeds)d):
        for gen in (random_cases.random_hyper_case, random_cases.random_case):
            c = gen(seed)
            if c['study'] == 'Study':
                continue
            S = cases.build(bl, c)
            S.communicator = GlooCommunicator()
            with np.errstate(all='ignore'):
                S.fit(**cases.fit_kwargs(c))
                want = oa.run(c)
            res = dict(logEvidence=S.logEvidence, localEvidence=S.localEvidence)
            gold = dict(logEvidence=want['logEvidence'], localEvidence=want['localEvidence'])
            if len(np.atleast_1d(S.logEvidenceList)) > 1:
                res.update(logEvidenceList=np.array(S.logEvidenceList), hyperParameterDistribution=S.hyperParameterDistribution)
                gold.update(logEvidenceList=np.asarray(want['logEvidenceList']), hyperParameterDistribution=np.asarray(want['hyperParameterDistribution']))
                if not np.all(np.isfinite(gold['logEvidenceList'])):
                    res['localEvidence'] = gold['localEvidence']      # np.empty left-overs of stopped chains in the reference
            flags = cases.fit_kwargs(c)
            if not flags.get('evidenceOnly') and np.isfinite(want['logEvidence']) and want.get('posteriorSequence') is not None:
                res['posteriorMeanValues'] = S.posteriorMeanValues
                gold['posteriorMeanValues'] = np.asarray(want['posteriorMeanValues'])
                if dist.get_rank() == 0:
                    res['posteriorSequence'] = S.posteriorSequence
                    gold['posteriorSequence'] = np.asarray(want['posteriorSequence'])
                else:
                    assert S.posteriorSequence is None
            compare.check(res, gold, dict(compare.ORACLE_TOL, post_rtol=1e-10, small_rtol=1e-10))
            done += 1
    print('rank', dist.get_rank(), 'random hyper-studies ok:', done)
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_random_hyperstudies_match_oracle(tmp_path, world):
    """Seeded random hyper- and change-point studies (tests/random_cases.py: hyper-priors, two hyper-parameters, serial models,
    stopped chains, evidenceOnly / forwardOnly) dealt out to 2 and 3 ranks: gather + accumulator merge = the unsharded oracle."""
    pytest.importorskip('torch')
    script = tmp_path / 'worker.py'
    script.write_text(RANDOM_WORKER % dict(root=ROOT, port=free_port(), world=world, seeds=27))
    env = dict(os.environ, OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, out[-3000:])
        assert 'random hyper-studies ok:' in out, out[-2000:]
