"""Turns a case description (tests/cases.py) into calls of the CPU oracle (oracle/bl_oracle.py)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import bl_oracle as orc   # noqa: E402
import cases                           # noqa: E402

OM_NAME = {'Poisson': 'poisson', 'Gaussian': 'gaussian', 'GaussianMean': 'gaussian_mean', 'Bernoulli': 'bernoulli',
           'Laplace': 'laplace', 'WhiteNoise': 'white_noise', 'AR1': 'ar1', 'ScaledAR1': 'scaled_ar1'}


class _Orc:
    """namespace so that cases.make_values can resolve cint/oint against the oracle's helpers"""
    cint = staticmethod(orc.cint)
    oint = staticmethod(orc.oint)


def estimate_values(om, pname_index, raw):
    """Default grids (observationModels.py:516-518 Poisson; :581-587 Gaussian; :720-726 GaussianMean)."""
    if om == 'poisson':
        return orc.oint(0, 1.25 * np.nanmax(np.ravel(raw)), 1000)
    if om == 'gaussian':
        mean, std = np.nanmean(np.ravel(raw)), np.nanstd(np.ravel(raw))
        return orc.cint(mean - 2 * std, mean + 2 * std, 200) if pname_index == 0 else orc.oint(0, 2 * std, 200)
    if om == 'gaussian_mean':
        obs = np.array([d[0] for d in raw])
        lo, hi = np.nanmin(obs), np.nanmax(obs)
        return orc.oint(lo - (hi - lo), hi + (hi - lo), 1000)
    if om == 'laplace':                                  # observationModels.py:637-652
        mean, std = np.nanmean(np.ravel(raw)), np.nanstd(np.ravel(raw))
        return orc.cint(mean - 2 * std, mean + 2 * std, 200) if pname_index == 0 else orc.oint(0, np.sqrt(2) * std, 200)
    raise ValueError(om)


def flatten_tm(spec, param_names):
    """-> ops, hyper values (scalar/array/'all'), hyper priors; in the order of the reference's flattened
    hyper-parameter list (core.py:623-656)."""
    kind = spec[0]
    if kind == 'Static':
        return [('static',)], [], []
    if kind == 'GRW':
        return [('grw', param_names.index(spec[3]))], [cases.make_values(_Orc, spec[2])], [cases.make_prior(spec[4])]
    if kind == 'ChangePoint':
        return [('changepoint',)], [cases.make_values(_Orc, spec[2])], [cases.make_prior(spec[3])]
    if kind == 'Combined':
        ops, vals, pri = [], [], []
        for s in spec[1]:
            o, v, p = flatten_tm(s, param_names)
            ops += o
            vals += v
            pri += p
        return ops, vals, pri
    if kind == 'RS':
        return [('regimeswitch',)], [cases.make_values(_Orc, spec[2])], [cases.make_prior(spec[3])]
    if kind == 'Independent':
        return [('independent',)], [], []
    if kind == 'Deterministic':
        import inspect
        fn = cases.FUNCS[spec[1]]
        sp = inspect.getfullargspec(fn)
        names = list(sp.args[1:])
        vals = [np.array(dflt) if isinstance(dflt, (list, tuple)) else dflt for dflt in sp.defaults]
        ops = [('deterministic', param_names.index(spec[2]), -1, 0, fn, names)] + [('deterministic_arg',)] * (len(names) - 1)
        return ops, vals, [None] * len(names)
    if kind == 'AlphaStable':
        return ([('alphastable', param_names.index(spec[5])), ('alphastable_arg',)],
                [cases.make_values(_Orc, spec[2]), cases.make_values(_Orc, spec[4])], [None, None])
    if kind == 'Bivariate':
        return ([('bivariate',), ('bivariate_arg',), ('bivariate_arg',)],
                [cases.make_values(_Orc, spec[2]), cases.make_values(_Orc, spec[4]), cases.make_values(_Orc, spec[6])], [None, None, None])
    if kind == 'NE':
        return [('notequal',)], [cases.make_values(_Orc, spec[2])], [cases.make_prior(spec[3])]
    if kind == 'Serial':
        # sub-models first (tagged with their segment), then the serial model's own break-/change-points
        ops, vals, pri, seg = [], [], [], 0
        bops, bvals, bpri = [], [], []
        for s in spec[1]:
            if s[0] == 'BreakPoint':
                bops.append(('breakpoint', None, -1, 0)); bvals.append(cases.make_values(_Orc, s[2])); bpri.append(cases.make_prior(s[3]))
            elif s[0] == 'ChangePoint':
                bops.append(('changepoint', None, -1, 1)); bvals.append(cases.make_values(_Orc, s[2])); bpri.append(cases.make_prior(s[3]))
            else:
                o, v, p = flatten_tm(s, param_names)
                ops += [(x[0], x[1] if len(x) > 1 else None, seg, 0) + tuple(x[4:]) for x in o]
                vals += v
                pri += p
                seg += 1
        return ops + bops, vals + bvals, pri + bpri
    raise ValueError(spec)


def run(case):
    c = cases.CASES[case] if isinstance(case, str) else case
    raw = cases.make_data(c['data'])
    om_cls, params, prior_spec = c['om']
    scipy_rv = None
    if om_cls.startswith('SciPy:'):           # plug-in model: the distribution's own pdf, evaluated here (observationModels.py:146-269)
        import scipy.stats
        scipy_rv = getattr(scipy.stats, om_cls.split(':')[1])
        om = 'table'
    else:
        om = OM_NAME[om_cls]
    marginals = []
    for k, (pname, values) in enumerate(params):
        v = cases.make_values(_Orc, values)
        marginals.append(estimate_values(om, k, raw) if v is None else v)
    g = orc.Grid(marginals)
    pnames = [p[0] for p in params]

    seg = 1 if scipy_rv is not None else orc.OM_INFO[om][0]
    data = orc.moving_window(raw, seg)
    ts = c.get('timestamps')
    ts = np.arange(len(raw)) if ts is None else np.asarray(ts)
    ts = ts[seg - 1:]

    lik_table = None
    if scipy_rv is not None:
        # ObservationModel.processedPdf (observationModels.py:35-56): product over the data dimensions; a NaN anywhere in the
        # segment leaves the step without information
        def _lik(segm):
            segm = np.asarray(segm, dtype=float)
            if np.any(np.isnan(segm)):
                return np.ones(g.size)
            kw_rv = dict(zip(pnames, g.grid))
            if segm.ndim == 2:
                out = np.ones(g.size)
                for col in segm.T:
                    out = out * scipy_rv.pdf(col[0], **kw_rv)
                return out
            return scipy_rv.pdf(segm[0], **kw_rv) * np.ones(g.size)
        with np.errstate(all='ignore'):
            lik_table = np.array([_lik(d) for d in data])
    if scipy_rv is not None:
        prior_obj = None if prior_spec == 'default' else cases.make_prior(prior_spec)       # SciPy models: flat prior by default
    else:
        prior_obj = orc.jeffreys(om) if prior_spec == 'default' else cases.make_prior(prior_spec)
    prior = orc.compute_prior(g, prior_obj)
    reset = orc.changepoint_prior(g, prior_obj)
    indep = reset / np.prod(g.lattice)

    ops, vals, hpriors = flatten_tm(c['tm'], pnames)
    kw = dict(c.get('fit', {}))
    fo, eo = kw.get('forwardOnly', False), kw.get('evidenceOnly', False)

    if c['study'] == 'Study':
        r = orc.fit(g, om, data, ts, prior, ops, orc.align_values(ops, vals), forward_only=fo, evidence_only=eo,
                    reset=reset, indep=indep, lik_table=lik_table)
        r['prior'] = prior
        r['grid'] = g
        return r

    vals = [ts[:-1] if (isinstance(v, str) and v == 'all') else v for v in vals]
    if len(vals) == 0 or np.prod([np.size(v) for v in vals]) <= 1:
        # <= 1 hyper-grid point: falls back to Study.fit (core.py:1434-1441)
        r = orc.fit(g, om, data, ts, prior, ops, orc.align_values(ops, [np.ravel(v)[0] for v in vals]),
                    forward_only=fo, evidence_only=eo, reset=reset, indep=indep, lik_table=lik_table)
        r['prior'] = prior
        r['grid'] = g
        return r
    hv, pv, const = orc.hyper_grid(vals, hpriors)
    extra = {}
    if c['study'] == 'ChangepointStudy':
        hops = [o for o in ops if o[0] not in ('static', 'independent')]
        cols = [k for k, op in enumerate(hops) if op[0] == 'changepoint' and not orc._is_boundary(op)]
        if not cols:        # break-/change-points of a serial model (reference core.py:1793-1815)
            cols = [k for k, op in enumerate(hops) if orc._is_boundary(op)]
        mask, hv_m, pv_m = orc.changepoint_mask(hv, pv, cols)
        extra = dict(allHyperGridValues=hv, mask=mask)
        hv, pv = hv_m, pv_m
    r = orc.hyper_fit(g, om, data, ts, prior, ops, hv, pv, const, forward_only=fo, evidence_only=eo, reset=reset,
                      n_jobs=kw.get('nJobs', 1), indep=indep, lik_table=lik_table)
    if c['study'] == 'ChangepointStudy':                       # core.py:1846-1852
        temp = np.zeros(len(extra['mask']))
        temp[extra['mask']] = r['hyperParameterDistribution']
        r['hyperParameterDistribution'] = temp
        temp = np.zeros(len(extra['mask']))
        temp[extra['mask']] = pv
        pv_out = temp
    else:
        pv_out = pv
    r.update(extra)
    r.update(prior=prior, grid=g, hyperGridValues=hv, flatHyperPriorValues=pv_out, hyperGridConstant=const)
    return r


def load_golden(case):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', case + '.npz')
    return dict(np.load(path))


def run_online(case):
    """The oracle's restatement of OnlineStudy.step over an ONLINE_CASES entry -> per-step results like the golden file."""
    c = cases.ONLINE_CASES[case] if isinstance(case, str) else case
    om_cls, params, prior_spec = c['om']
    scipy_rv = None
    if om_cls.startswith('SciPy:'):           # plug-in model (any number of parameters): the distribution's own pdf, flat prior by default
        import scipy.stats
        scipy_rv = getattr(scipy.stats, om_cls.split(':')[1])
        om = 'table'
    else:
        om = OM_NAME[om_cls]
    g = orc.Grid([cases.make_values(_Orc, v) for _, v in params])
    pnames = [p[0] for p in params]
    if scipy_rv is not None:
        prior_obj = None if prior_spec == 'default' else cases.make_prior(prior_spec)
    else:
        prior_obj = orc.jeffreys(om) if prior_spec == 'default' else cases.make_prior(prior_spec)
    prior = orc.compute_prior(g, prior_obj)
    reset = orc.changepoint_prior(g, prior_obj)
    specs = c.get('models') or [('transition model', c['set_tm'])]
    models = []
    for name, spec in specs:
        ops, vals, pri = flatten_tm(spec, pnames)
        if len(vals) == 0:
            models.append(dict(ops=ops, values=[orc.align_values(ops, [])], prior_values=np.array([1.]), grid_constants=[],
                               reset=reset, indep=reset / np.prod(g.lattice)))
            continue
        pri2 = []
        for p, v in zip(pri, vals):
            if p is not None and not hasattr(p, '__call__') and not isinstance(p, (list, tuple, np.ndarray)):
                from sympy import lambdify, abc                       # SymPy random variable (test_onlinestudy.py:41)
                from sympy.stats import density
                p = ('density', lambdify([abc.x], density(p)(abc.x), modules=['numpy'])(np.asarray(v, dtype=float)))
            pri2.append(p)
        hv, pv, const = orc.hyper_grid(vals, pri2)
        models.append(dict(ops=ops, values=[orc.align_values(ops, row) for row in hv], prior_values=pv, grid_constants=const,
                           reset=reset, indep=reset / np.prod(g.lattice)))
    state = orc.OnlineState(g, prior, models, c.get('tm_prior'))
    seg = 1 if scipy_rv is not None else orc.OM_INFO[om][0]
    raw = np.asarray(cases.online_data(c), dtype=float)
    out = dict(posteriorSequence=[], posteriorMeanValues=[], transitionModelSequence=[], localTransitionModelSequence=[],
               hyperParameterSequence=[])
    for k in range(len(raw)):
        if k + 1 < seg:
            continue
        with np.errstate(all='ignore'):
            lik = None
            if scipy_rv is not None:           # ObservationModel.processedPdf (observationModels.py:35-56): a NaN leaves the step without information
                x = raw[k]
                lik = np.ones(g.size) if np.isnan(x) else scipy_rv.pdf(x, **dict(zip(pnames, g.grid))) * np.ones(g.size)
            r = orc.online_step(state, om, raw[k + 1 - seg:k + 1], lik=lik)
        out['posteriorSequence'].append(r['marginalizedPosterior'])
        out['posteriorMeanValues'].append(r['posteriorMeanValues'])
        out['transitionModelSequence'].append(r['transitionModelDistribution'])
        out['localTransitionModelSequence'].append(r['localTransitionModelDistribution'])
        out['hyperParameterSequence'].append(r['hyperParameterDistribution'])
    out['logEvidence'] = state.log_evidence
    out['parameterPosterior'] = state.parameter_posterior
    out['logEvidenceList'] = state.log_evidence_list
    return out
