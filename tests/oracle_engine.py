"""
TEST DOUBLE: an object with the interface of bayesloop_amd.engine.HipEngine whose arithmetic is the CPU oracle.

It exists so that the host-side logic of the product (grid / prior construction, transition-program compilation,
hyper-grid plumbing, sharding and the merge of bayesloop_amd.dist) can be tested on a machine without a GPU
(``-m "not gpu"``).  It is installed with ``bayesloop_amd.set_engine`` by tests only; the product never imports it.
"""
from __future__ import annotations

import numpy as np

from oracle import bl_oracle as orc
from bayesloop_amd import _abi
from bayesloop_amd.engine import FitResult

OM = {_abi.OM_POISSON: 'poisson', _abi.OM_GAUSSIAN: 'gaussian', _abi.OM_GAUSSIAN_MEAN: 'gaussian_mean',
      _abi.OM_TABLE: 'table', _abi.OM_BERNOULLI: 'bernoulli', _abi.OM_LAPLACE: 'laplace', _abi.OM_WHITE_NOISE: 'white_noise',
      _abi.OM_AR1: 'ar1', _abi.OM_SCALED_AR1: 'scaled_ar1'}
OPS = {_abi.OP_STATIC: 'static', _abi.OP_GRW: 'grw', _abi.OP_CHANGEPOINT: 'changepoint', _abi.OP_REGIMESWITCH: 'regimeswitch',
       _abi.OP_INDEPENDENT: 'independent', _abi.OP_BREAKPOINT: 'breakpoint', _abi.OP_NOTEQUAL: 'notequal',
       _abi.OP_BIVARIATE: 'bivariate', _abi.OP_BIVARIATE_ARG: 'bivariate_arg',
       _abi.OP_ALPHASTABLE: 'alphastable', _abi.OP_ALPHASTABLE_ARG: 'alphastable_arg', _abi.OP_DETERMINISTIC: 'shift_table'}


class OracleEngine:
    name = 'oracle-test-double'

    def __init__(self):
        self._post = None
        self._posterior_owner = None
        self.acc_log = None
        self.acc_lin = None
        self.fits = 0
        self._carry = {}
        self._mix = None

    def _unpack(self, p):
        g = orc.Grid(p.marginal)
        ops = [(OPS[op[0]], op[1], op[2] if len(op) > 2 else -1, op[3] if len(op) > 3 else 0) for op in p.ops
               if op[0] != _abi.OP_DETERMINISTIC_ARG]
        # (a DETERMINISTIC op is followed by 2 T ARG ops = its table of shifts: folded into one oracle op, see _values)
        self._abi_ops = list(p.ops)
        self._ts = np.asarray(p.timestamps, dtype=float)
        data = np.asarray(p.data, dtype=float)
        lik = None if p.lik is None else np.asarray(p.lik, dtype=float).reshape([p.T] + g.size)
        reset = None if p.reset_prior is None else np.asarray(p.reset_prior, dtype=float).reshape(g.size)
        self._indep = None if p.indep_prior is None else np.asarray(p.indep_prior, dtype=float).reshape(g.size)
        return g, OM[p.obs_model], ops, data, lik, reset

    def _values(self, row):
        """one chain's ABI op values -> the oracle's value list (None for ops without a value; a DETERMINISTIC op takes the
        2 T values of the ARG ops behind it as its shift table)"""
        vals, T = [], len(self._ts)
        for k, op in enumerate(self._abi_ops):
            if op[0] == _abi.OP_DETERMINISTIC_ARG:
                continue
            if op[0] == _abi.OP_DETERMINISTIC:
                vals.append((self._ts, np.asarray(row[k + 1:k + 1 + 2 * T], dtype=float)))
            elif op[0] in (_abi.OP_STATIC, _abi.OP_INDEPENDENT):
                vals.append(None)
            else:
                vals.append(row[k])
        return vals

    def _online(self, problem, op_values, resume, carry):
        """One forward step from the prior or from the carried states (OnlineStudy.step, core.py:2157-2174)."""
        g, om, ops, data, lik, reset = self._unpack(problem)
        assert problem.T == 1
        n = len(op_values)
        L = orc.processed_pdf(om, g.grid, data[0]) if lik is None else lik[0]
        logE, local = np.zeros(n), np.zeros((n, 1))
        states = np.zeros([n] + g.size)
        for c in range(n):
            vals = self._values(op_values[c])
            with np.errstate(all='ignore'):
                if resume:
                    alpha = orc.transition_forward(ops, vals, self._carry[problem.carry_slot][c], problem.resume_time, g, reset,
                                                   self._indep) * L
                else:
                    alpha = np.asarray(problem.prior, dtype=float).reshape(g.size) * L
                ni = np.sum(alpha)
                local[c, 0] = ni * g.dV
                logE[c] = np.log(ni) + np.log(g.dV)
                states[c] = alpha / ni
        if carry:
            self._carry[problem.carry_slot] = states
        return FitResult(logE, local, None, np.full(n, -1, dtype=np.int64), np.zeros(n, dtype=np.int32), {})

    def carry_mix(self, slot, weights, accumulate=False):
        st = self._carry[slot]
        m = np.tensordot(np.asarray(weights, dtype=float), st, axes=(0, 0))
        self._mix = m if not accumulate else self._mix + m

    def carry_read(self, slot, chain, grid_size):
        return (self._mix if chain < 0 else self._carry[slot][chain]).copy().reshape(grid_size)

    def carry_write(self, slot, states):
        self._carry[slot] = np.array(states, dtype=float)

    def carry_release(self, slot=-1):
        if slot < 0:
            self._carry.clear()
        else:
            self._carry.pop(slot, None)

    def fit(self, problem, op_values, forward_only=False, evidence_only=False, keep_posterior=False,
            accumulate=False, log_chain_weight=None, owner=None, resume=False, carry=False):
        if resume or carry:
            return self._online(problem, op_values, resume, carry)
        if self._posterior_owner is not None and self._posterior_owner is not owner:
            prev, self._posterior_owner = self._posterior_owner, None
            prev._materialize_posterior()
        g, om, ops, data, lik, reset = self._unpack(problem)
        self._gs = list(g.size)
        n, T, ndim = len(op_values), problem.T, len(g.size)
        logE = np.zeros(n)
        local = np.zeros((n, T))
        means = None if evidence_only else np.zeros((n, ndim, T))
        astep = np.full(n, -1, dtype=np.int64)
        aphase = np.zeros(n, dtype=np.int32)
        self._post = None
        for c in range(n):
            vals = self._values(op_values[c])
            with np.errstate(all='ignore'):
                r = orc.fit(g, om, data, problem.timestamps, np.asarray(problem.prior).reshape(g.size), ops, vals,
                            forward_only=forward_only, evidence_only=evidence_only, reset=reset, lik_table=lik,
                            indep=self._indep, beta_init=getattr(problem, 'backward_init', None))
            self.fits += 1
            logE[c] = r['logEvidence']
            local[c] = r['localEvidence']
            if r['abort'] is not None:
                astep[c] = r['abort'][1]
                aphase[c] = 0 if r['abort'][0] == 'forward' else 1
                continue
            if not evidence_only:
                means[c] = r['posteriorMeanValues']
                if keep_posterior:
                    self._post = r['posteriorSequence'].copy()
                if accumulate and np.isfinite(r['logEvidence']):
                    with np.errstate(divide='ignore'):
                        flat = dict(r, posteriorSequence=r['posteriorSequence'].reshape(T, -1))
                        self.acc_log = orc.hyper_accumulate(self.acc_log, flat, log_chain_weight[c])
        if keep_posterior and not evidence_only:
            self._posterior_owner = owner
        return FitResult(logE, local, means, astep, aphase, {})

    def posterior(self, chain, T, grid_size, t0=0, t1=None):
        return self._post.reshape([T] + list(grid_size))[t0:t1]

    def marginal(self, source, chain, keep_axis, T, n_keep):
        post = self._post.reshape([T] + self._gs) if source == 0 else self.acc_final.reshape([T] + self._gs)
        axes = tuple(a + 1 for a in range(len(self._gs)) if a != keep_axis)
        return post.sum(axis=axes) if axes else post.copy()

    def time_average(self, source, chain, grid_size):
        post = self._post if source == 0 else self.acc_final
        return post.reshape([-1] + list(grid_size)).mean(axis=0)

    def release_posterior(self, owner=None):
        self._posterior_owner = None

    def last_timing(self):
        return {}

    # ---- accumulator (log space inside, linear space for the cross-rank merge) -----------------------------------
    def accum_begin(self, T, G, owner=None):
        prev = getattr(self, '_accum_owner', None)          # as HipEngine.accum_begin: the previous study's average posterior
        if prev is not None and prev is not owner:          # still lives in the accumulator that is about to be reused
            self._accum_owner = None
            prev._materialize_posterior()
        self._accum_owner = owner
        self.acc_shape = (T, G)
        self.acc_log = np.zeros((T, G)) - np.inf
        self.acc_lin = None

    def accum_fold_host(self, posterior, log_weight):
        """as HipEngine.accum_fold_host: one chain whose posterior sequence the caller holds (log_weight = logEvidence + log prior value)"""
        if np.isfinite(log_weight):
            post = np.array(posterior, dtype=float).reshape(self.acc_log.shape)
            self.acc_log = orc.hyper_accumulate(self.acc_log, dict(posteriorSequence=post, logEvidence=0.0), log_weight)

    def accum_row_stats(self, problem):
        """as HipEngine.accum_row_stats: per-step sums of exp(acc_log - own reference exponent)"""
        T, G = self.acc_shape
        g = orc.Grid(problem.marginal)
        out = np.zeros((T, 1 + len(g.size)))
        ref = np.amax(self.acc_log)
        if self.acc_lin is not None or np.isfinite(ref):
            # (after accum_rescale -- and the cross-rank merge -- the linear accumulator is what the device holds)
            lin = np.array(self.acc_lin, dtype=float).reshape(T, G) if self.acc_lin is not None else np.exp(self.acc_log - ref)
            out[:, 0] = lin.sum(axis=1)
            for k in range(len(g.size)):
                out[:, 1 + k] = lin @ g.grid[k].ravel()
        return out

    def _acc3(self, problem):
        return [problem.T] + list(problem.grid_size)

    def fit_shape(self, a, shape):
        return a.reshape(shape)

    def accum_log_ref(self):
        return float(np.amax(self.acc_log)), 0

    def accum_rescale(self, new_log_ref):
        self.acc_lin = np.ascontiguousarray(np.exp(self.acc_log - new_log_ref).ravel())

    def accum_finalize(self, problem):
        T, G = self.acc_shape
        self._gs = list(problem.grid_size)
        if self.acc_lin is not None:
            avg = np.array(self.acc_lin, dtype=float).reshape(T, G)
        else:
            avg = np.exp(self.acc_log - np.amax(self.acc_log))
        avg /= avg.sum(axis=1)[:, None]
        self.acc_final = avg
        g = orc.Grid(problem.marginal)
        means = np.empty((len(g.size), T))
        for k in range(len(g.size)):
            means[k] = np.array([np.sum(p.reshape(g.size) * g.grid[k]) for p in avg])
        return means

    def accum_read(self, T, grid_size, t0=0, t1=None):
        return self.acc_final.reshape([T] + list(grid_size))[t0:t1].copy()

    def accum_end(self):
        pass

    def synchronize(self):
        pass

    # ---- what bayesloop_amd.dist.LocalGroup asks of an engine (HyperStudy.fit(nJobs=N): several engines in one process) ----
    def accum_shape(self):
        return self.acc_shape

    def accum_peer_reduce(self, others, row0, row1):
        T, G = self.acc_shape
        mine = self.acc_lin.reshape(T, G)
        for o in others:
            mine[row0:row1] += o.acc_lin.reshape(T, G)[row0:row1]

    def accum_peer_gather(self, others, bounds):
        T, G = self.acc_shape
        mine = self.acc_lin.reshape(T, G)
        for o, (a, b) in zip(others, bounds):
            mine[a:b] = o.acc_lin.reshape(T, G)[a:b]
