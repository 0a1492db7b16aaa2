"""
Study classes -- ``bl.Study``, ``bl.HyperStudy``, ``bl.ChangepointStudy`` -- with the reference's public surface
(bayesloop/core.py:39-486, 1118-1495, 1743-1852) and a ``fit()`` that runs on an MI355X.

What stays on the host (cheap, data-independent of the grid size): building the parameter grid and the prior, formatting
the data, compiling the transition model into a flat program, creating the hyper-parameter grid and hyper-priors, and
the final evidence algebra over the handful of per-chain scalars.  Everything that touches a grid-sized array -- the
likelihood, the alpha/beta recursion, the normalisers, the posterior means and the evidence-weighted average over
hyper-parameter values -- runs in the HIP kernels behind :mod:`bayesloop_amd.engine`.
"""
from __future__ import annotations

from collections.abc import Iterable

import numpy as np

from . import _abi
from . import engine as _engine_mod
from .engine import FitProblem, DevicePosterior
from .exceptions import ConfigurationError, PostProcessingError
from .helper import flatten
from .observationModels import ObservationModel, device_code
from .preprocessing import movingWindow
from . import transitionModels as _tm_mod
from .transitionModels import (TransitionModel, ChangePoint, CombinedTransitionModel, SerialTransitionModel,
                               BivariateRandomWalk, AlphaStableRandomWalk, Deterministic)

COAL_MINING = (5, 4, 1, 0, 4, 3, 4, 0, 6, 3, 3, 4, 0, 2, 6, 3, 3, 5, 4, 5, 3, 1, 4, 4, 1, 5, 5, 3, 4, 2, 5, 2, 2, 3, 4, 2,
               1, 3, 2, 2, 1, 1, 1, 1, 3, 0, 0, 1, 0, 1, 1, 0, 0, 3, 1, 0, 3, 2, 2, 0, 1, 1, 1, 0, 1, 0, 1, 0, 0, 0, 2, 1,
               0, 0, 0, 1, 1, 0, 2, 3, 3, 1, 1, 2, 1, 1, 1, 1, 2, 3, 3, 0, 0, 0, 1, 4, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0,
               1, 0)   # UK coal mining disasters per year, 1852-1961 (the reference's example data, core.py:82-88)


def _hyper_slots(model):
    """[(model, index, name)] of all hyper-parameters of a (nested) transition model in the reference's flattened order:
    sub-models first (depth first), then the model's own hyper-parameters (reference core.py:623-647)."""
    out = []
    for m in getattr(model, 'models', []):
        out += _hyper_slots(m)
    for k, name in enumerate(getattr(model, 'hyperParameterNames', [])):
        out.append((model, k, name))
    return out


def _logsumexp(a):
    """scipy.special.logsumexp of a 1-D array, including its convention for a non-finite maximum (all -inf -> -inf)."""
    a = np.asarray(a, dtype=float)
    m = np.amax(a)
    if not np.isfinite(m):
        m = 0.0
    with np.errstate(divide='ignore', over='ignore', invalid='ignore'):
        return m + np.log(np.sum(np.exp(a - m)))


def _plots(kind):
    """Adds the reference's ``plot=True`` behaviour to an accessor: the numbers come from the undecorated method, the figure
    from :mod:`bayesloop_amd.plotting` (matplotlib is only imported when something is plotted)."""
    def wrap(fn):
        import functools

        import inspect
        sig = inspect.signature(fn)
        own = [n for n, q in sig.parameters.items() if q.kind in (q.POSITIONAL_OR_KEYWORD, q.KEYWORD_ONLY)]

        @functools.wraps(fn)
        def accessor(self, *args, **kwargs):
            # the accessor's own parameters (t, name, names, density, ...) may arrive by keyword, as in the reference; only
            # what is left over is matplotlib styling
            style = {k: kwargs.pop(k) for k in list(kwargs) if k not in own}
            bound = sig.bind(self, *args, **kwargs)
            bound.apply_defaults()
            a = bound.arguments
            plot = a.get('plot', False)
            figure, subplot = a.get('figure', None), a.get('subplot', 111)
            density = a.get('density', True)
            call = {k: v for k, v in a.items() if k not in ('self', 'kwargs')}
            call['plot'] = False
            out = fn(self, **call)
            if not plot:
                return out
            from . import plotting
            name = a.get('name', a.get('names'))
            if kind == 'param_dist':
                plotting.distribution(out[0], out[1], name, density=density, **style)
            elif kind == 'param_dists':
                plotting.marginal_image(self.formattedTimestamps, self.boundaries[self._parameterIndex(name)], out[1],
                                        color=style.get('c', style.get('color', 'b')))
            elif kind == 'hyper_dist':
                plotting.bars(out[0], out[1], name, **style)
            elif kind == 'joint':
                k = [self._getHyperParameterIndex(self.transitionModel, n) for n in name]
                plotting.joint_bars(out[0], out[1], out[2], list(name), [self.hyperGridConstant[k[0]], self.hyperGridConstant[k[1]]],
                                    figure=figure, subplot=subplot, **style)
            elif kind == 'duration':
                plotting.bars(out[0], out[1], 'duration between {} and {} (in time steps)'.format(*name), width=out[0][0], **style)
            return out
        return accessor
    return wrap


class Study(object):
    """Fit with fixed hyper-parameter values (reference core.py:39-486)."""

    def __init__(self, silent=False):
        self.observationModel = None
        self.transitionModel = None
        self.gridSize = []
        self.boundaries = []
        self.marginalGrid = []
        self.grid = []
        self.latticeConstant = []
        self.rawData = np.array([])
        self.formattedData = np.array([])
        self.rawTimestamps = None
        self.formattedTimestamps = None
        self._posteriorSequence = []
        self._posterior_pending = None
        self.posteriorMeanValues = []
        self.logEvidence = 0
        self.localEvidence = []
        self.selectedHyperParameters = []
        self.fitWarningCounter = 0
        self.lastTiming = {}
        if not silent:
            print('+ Created new study.')

    # ---- results that may still live on the GPU ------------------------------------------------------------------
    @property
    def posteriorSequence(self):
        self._materialize_posterior()
        return self._posteriorSequence

    @posteriorSequence.setter
    def posteriorSequence(self, value):
        self._posterior_pending = None
        self._posteriorSequence = value

    def _materialize_posterior(self):
        pending = self._posterior_pending
        if pending is not None:
            self._posterior_pending = None
            self._posteriorSequence = pending()

    def __getstate__(self):
        self._materialize_posterior()     # keep study objects picklable (reference fileIO.py:10-37)
        state = dict(self.__dict__)
        state['_posterior_pending'] = None
        return state

    @property
    def log10Evidence(self):
        return self.logEvidence / np.log(10)

    # ---- data ----------------------------------------------------------------------------------------------------
    def loadExampleData(self, silent=False):
        self.rawData = np.array(COAL_MINING)
        self.rawTimestamps = np.arange(1852, 1962)
        if not silent:
            print('+ Successfully imported example data.')

    def loadData(self, array, timestamps=None, silent=False):
        if isinstance(array, np.ndarray):
            self.rawData = array
        elif isinstance(array, list):
            if not silent:
                print('! WARNING: Data supplied as list, not as Numpy array. Converting list to Numpy array '
                      '(dtype=float).')
            self.rawData = np.array(array, dtype=float)
        else:
            raise ConfigurationError('Data type not supported. Please provide data as Numpy array.')
        self.rawTimestamps = np.arange(len(self.rawData))
        if timestamps is not None:
            if len(timestamps) == len(array):
                self.rawTimestamps = np.array(timestamps)
            elif not silent:
                print('! WARNING: Number of timestamps does not match number of data points. Omitting timestamps.')
        if not silent:
            print('+ Successfully imported array.')

    def load(self, array, timestamps=None, silent=False):
        self.loadData(array, timestamps=timestamps, silent=silent)

    # ---- models --------------------------------------------------------------------------------------------------
    def setObservationModel(self, L, silent=False):
        """Sets the likelihood and builds the regular parameter grid (reference core.py:130-176)."""
        self.observationModel = L
        self.marginalGrid, self.gridSize, self.boundaries, self.latticeConstant = [], [], [], []
        for values, name in zip(L.parameterValues, L.parameterNames):
            if values is None:
                try:
                    values = L.estimateParameterValues(name, self.rawData)
                except Exception:
                    raise ConfigurationError('Could not estimate parameter values for "{}".'.format(name))
                print('+ Estimated parameter interval for "{}": [{}, {}] ({} values).'
                      .format(name, values[0], values[-1], len(values)))
            values = np.array(values, dtype=float)
            if values.ndim != 1 or len(values) < 2:
                raise ConfigurationError('Parameter "{}" needs at least two grid values.'.format(name))
            self.marginalGrid.append(values)
            self.gridSize.append(len(values))
            self.boundaries.append([values[0], values[-1]])
            if np.any(np.abs(np.diff(np.diff(values))) > 10 ** -10):
                print('! WARNING: Supplied parameter values for "{}" are not equally spaced. Assuming categorical '
                      'parameter.'.format(name))
                self.latticeConstant.append(1.)
            else:
                self.latticeConstant.append(np.abs(values[0] - values[1]))
        self.grid = [m for m in np.meshgrid(*self.marginalGrid, indexing='ij')]
        if self.transitionModel is not None:
            self.transitionModel.latticeConstant = self.latticeConstant
        if not silent:
            print('+ Observation model: {}. Parameter(s): {}'.format(L, L.parameterNames))

    def setOM(self, L, silent=False):
        self.setObservationModel(L, silent=silent)

    def setTransitionModel(self, T, silent=False):
        if str(T) == 'Break-point':
            raise ConfigurationError('The "BreakPoint" transition model can only be used with the '
                                     '"SerialTransitionModel" class.')
        self.transitionModel = T
        T.study = self
        T.latticeConstant = self.latticeConstant
        if not silent:
            print('+ Transition model: {}. Hyper-Parameter(s): {}'.format(T, self._unpackAllHyperParameters(values=False)))

    def setTM(self, T, silent=False):
        self.setTransitionModel(T, silent=silent)

    def set(self, *args, **kwargs):
        for key in kwargs:
            if key not in ['silent']:
                raise TypeError("set() got an unexpected keyword argument '{}'".format(key))
        silent = kwargs.pop('silent', False)
        om = tm = False
        for model in args:
            if isinstance(model, ObservationModel):
                if om:
                    raise ConfigurationError('More than one observation model supplied.')
                om = True
                self.setObservationModel(model, silent=silent)
            elif isinstance(model, TransitionModel):
                if tm:
                    raise ConfigurationError('More than one transition model supplied.')
                tm = True
                self.setTransitionModel(model, silent=silent)
            else:
                raise ConfigurationError('Expected observation model or transition model instance as first argument.')

    # ---- prior ---------------------------------------------------------------------------------------------------
    def _computePrior(self, silent=False):
        """Prior on the grid (see :meth:`_evaluatePrior`).  The array of a callable / SymPy / uniform prior is kept for as
        long as the same prior object is evaluated on the same grid: on a 2048 x 2048 grid evaluating the default Jeffreys
        prior in numpy takes longer (13 ms) than the 200 forward steps of the fit on the GPU (5.5 ms).  An ndarray prior is
        always re-read (it may have been modified in place).  ``self.cachePrior = False`` switches the cache off."""
        prior = self.observationModel.prior
        if isinstance(prior, np.ndarray) or not getattr(self, 'cachePrior', True):
            return self._evaluatePrior(silent)
        key = (id(self.observationModel), tuple((len(m), float(m[0]), float(m[-1])) for m in self.marginalGrid),
               tuple(float(c) for c in self.latticeConstant))
        cached = getattr(self, '_prior_cache', None)
        if cached is not None and cached[0] == key and cached[1] is prior and cached[2] is self.observationModel:
            return cached[3]                       # (read-only array)
        p = self._evaluatePrior(silent)
        p.setflags(write=False)
        self._prior_cache = (key, prior, self.observationModel, p)
        return p

    def _evaluatePrior(self, silent=False):
        """Prior on the grid for None / ndarray / callable priors (reference core.py:184-235).  Unlike the reference
        an ndarray prior is copied, never normalised in place."""
        prior = self.observationModel.prior
        cell = np.prod(self.latticeConstant)
        if prior is None:
            if not silent:
                print('    + Set uniform prior with parameter boundaries.')
            p = np.ones(self.gridSize)
            p /= np.sum(p)
            p /= cell
            return p
        if isinstance(prior, np.ndarray):
            if tuple(prior.shape) != tuple(self.gridSize):
                raise ConfigurationError('Prior array does not match parameter grid size.')
            p = np.array(prior, dtype=float)
            norm = np.sum(p)
            if norm != 1.:
                p /= norm
                p /= cell
            if not silent:
                print('    + Set prior (numpy array).')
            return p
        if hasattr(prior, '__call__'):
            p = prior(*self.grid) * np.ones(self.gridSize)
            norm = np.sum(p)
            if norm != 1.:
                p /= norm
                p /= cell
            if not silent:
                print('    + Set prior (function): {}'.format(getattr(prior, '__name__', 'callable')))
            return p
        return self._sympy_prior(prior, silent)

    def _sympy_prior(self, prior, silent):
        """SymPy random variable(s) as prior: product density evaluated on the grid, not re-normalised
        (reference core.py:238-265)."""
        try:
            import sympy
            from sympy import lambdify, symbols, abc
            from sympy.stats import density
        except ImportError:
            raise ConfigurationError('Only None, arrays, functions or SymPy random variables can be used as a prior.')
        if type(prior) is sympy.stats.rv.RandomSymbol:
            prior = [prior]
        if not isinstance(prior, (list, tuple)):
            raise ConfigurationError('Only None, arrays, functions or SymPy random variables can be used as a prior.')
        if len(prior) != len(self.observationModel.parameterNames):
            raise ConfigurationError('Observation model contains {} parameters, but {} priors were provided.'
                                     .format(len(self.observationModel.parameterNames), len(prior)))
        import string
        x = symbols(' '.join(list(string.ascii_lowercase)[:len(prior)])) if len(prior) > 1 else [abc.x]
        pdf = 1
        for k, rv in enumerate(prior):
            if type(rv) is not sympy.stats.rv.RandomSymbol:
                raise ConfigurationError('Only lambda functions or SymPy random variables can be used as a prior.')
            pdf = pdf * density(rv)(x[k])
        if not silent:
            print('    + Set prior (sympy): {}'.format(pdf))
        return np.asarray(lambdify(x, pdf, modules=['numpy'])(*self.grid), dtype=float) * np.ones(self.gridSize)

    def _changepointPrior(self):
        """The distribution a change-point restarts from (reference transitionModels.py:300-312)."""
        prior = self.observationModel.prior
        if hasattr(prior, '__call__'):
            p = prior(*self.grid) * np.ones(self.gridSize)
        elif isinstance(prior, np.ndarray):
            p = np.array(prior, dtype=float)
        else:
            p = np.ones(self.gridSize)
        p = p / np.sum(p)
        return p * np.prod(self.latticeConstant)

    # ---- hyper-parameter plumbing (reference core.py:623-824) -------------------------------------------------------
    def _hyperSlots(self):
        """[(model, index into model.hyperParameterValues, name)] in the reference's flattened order."""
        if self.transitionModel is None:
            return []
        return _hyper_slots(self.transitionModel)

    def _unpackAllHyperParameters(self, values=True):
        return [m.hyperParameterValues[k] if values else name for m, k, name in self._hyperSlots()]

    def _selectedSlots(self):
        slots = self._hyperSlots()
        if not self.selectedHyperParameters:
            return slots
        out, used = [], set()
        for name in self.selectedHyperParameters:
            hit = [i for i, s in enumerate(slots) if s[2] == name and i not in used]
            if not hit:
                raise ConfigurationError('Could not find any hyper-parameter named {}.'.format(name))
            used.add(hit[0])
            out.append(slots[hit[0]])
        return out

    def _unpackSelectedHyperParameters(self):
        return [m.hyperParameterValues[k] for m, k, _ in self._selectedSlots()]

    def _setAllHyperParameters(self, x):
        for (m, k, _), v in zip(self._hyperSlots(), list(x)):
            m.hyperParameterValues[k] = v

    def _setSelectedHyperParameters(self, x):
        for (m, k, _), v in zip(self._selectedSlots(), list(x)):
            m.hyperParameterValues[k] = v
        return 1

    def _unpackChangepointNames(self, transitionModel=None):
        """Names of stand-alone change-points (reference core.py:761-780)."""
        return [name for m, k, name in self._hyperSlots() if isinstance(m, ChangePoint)]

    def _unpackBreakpointNames(self, transitionModel=None):
        """Names of the break-/change-points of serial transition models (reference core.py:782-801)."""
        return [name for m, k, name in self._hyperSlots() if isinstance(m, SerialTransitionModel)]

    def _getHyperParameterIndex(self, transitionModel, name):
        names = self._unpackAllHyperParameters(values=False)
        if name not in names:
            raise PostProcessingError('Could not find any hyper-parameter with name: {}.'.format(name))
        return names.index(name)

    def eval(self, query, t=None, silent=False):
        """Probability of an (in-)equality of (hyper-)parameters, or the distribution of an arithmetic combination of them
        (reference core.py:604-621; see :class:`bayesloop_amd.parser.Parser`)."""
        from .parser import Parser
        return Parser(self)(query, t=t, silent=silent)

    def getHyperParameterValue(self, name):
        return self._unpackAllHyperParameters(values=True)[self._getHyperParameterIndex(self.transitionModel, name)]

    def _checkConsistency(self):
        """Raised before any device work starts (reference core.py:1098-1115)."""
        if len(self.rawData) == 0:
            raise ConfigurationError('No data loaded.')
        if not self.observationModel:
            raise ConfigurationError('No observation model chosen.')
        if not self.transitionModel:
            raise ConfigurationError('No transition model chosen.')
        names = self._unpackAllHyperParameters(values=False)
        u, i = np.unique(names, return_inverse=True)
        duplicates = u[np.bincount(i) > 1] if len(names) else []
        if len(duplicates) > 0:
            raise ConfigurationError('Detected duplicate hyper-parameter names: {}.'.format(duplicates))

    # ---- compiling the study into a device problem ---------------------------------------------------------------
    def _formatData(self):
        seg = self.observationModel.segmentLength
        self.formattedData = movingWindow(self.rawData, seg)
        self.formattedTimestamps = self.rawTimestamps[seg - 1:]

    def _compile(self, silent=True):
        """-> (FitProblem, program) ; program = [(kind, axis, model, hyper index)] in list order."""
        om = self.observationModel
        program = self.transitionModel._program(om.parameterNames)
        if not 1 <= len(self.gridSize) <= _abi.MAX_DIM:
            raise ConfigurationError('The MI355X engine supports observation models with 1 to {} parameters '
                                     '(got {}).'.format(_abi.MAX_DIM, len(self.gridSize)))
        if len(self.gridSize) > 2 and not all(op[0] in (_abi.OP_GRW, _abi.OP_STATIC, _abi.OP_CHANGEPOINT) and op[4] < 0
                                              for op in program):
            raise ConfigurationError('Observation models with more than two parameters can be combined with GaussianRandomWalk, '
                                     'Static and ChangePoint transition models.')
        prior = self._computePrior(silent=silent)
        reset = self._changepointPrior() if any(op[0] == _abi.OP_CHANGEPOINT for op in program) else None
        indep = None
        if any(op[0] == _abi.OP_INDEPENDENT for op in program):
            indep = self._changepointPrior() / np.prod(self.latticeConstant)      # sum 1 (reference transitionModels.py:351-360)
        data = np.asarray(self.formattedData, dtype=float)
        lik = None
        code = device_code(om)
        if code == _abi.OM_TABLE:
            # the model's own pdf, evaluated once per time step on the host (plug-in interface of the reference)
            lik = np.array([np.asarray(om.processedPdf(self.grid, seg), dtype=float) * np.ones(self.gridSize)
                            for seg in self.formattedData])
        program = self._expandProgram(program, len(data))
        problem = FitProblem(obs_model=code, marginal=self.marginalGrid, lattice=self.latticeConstant, data=data,
                             timestamps=np.asarray(self.formattedTimestamps, dtype=float), prior=prior,
                             ops=[(op[0], op[1], op[4], op[5]) for op in program], reset_prior=reset, indep_prior=indep,
                             lik=lik,
                             seg_len=om.segmentLength)
        return problem, program

    @staticmethod
    def _expandProgram(program, T):
        """A Deterministic op is followed by 2 T DETERMINISTIC_ARG ops carrying its per-step shifts (include/blhip.h); their
        hyper index is the pair ('shift', position) resolved by :meth:`_opValueMatrix`."""
        out = []
        for op in program:
            out.append(op)
            if op[0] == _abi.OP_DETERMINISTIC:
                out += [(_abi.OP_DETERMINISTIC_ARG, 0, op[2], ('shift', q), -1, 0) for q in range(2 * T)]
        return out

    def _opValueMatrix(self, program, hyper_rows=None, timestamps=None, resume_time=-1.0):
        """(n_chains, n_ops) op values.  ``hyper_rows``: (n_chains, n_hyper) values in the order of the flattened
        hyper-parameter list (None: one chain with the models' current values).  Ops without a value get NaN; the
        DETERMINISTIC_ARG ops get the shifts the Deterministic model's function gives for the chain's hyper-parameters."""
        slots = self._hyperSlots()
        if hyper_rows is None:
            row = []
            for m, k, name in slots:
                v = m.hyperParameterValues[k]
                if isinstance(v, str) or np.ndim(v) != 0:
                    raise ConfigurationError('Hyper-parameter "{}" holds several values; use a HyperStudy to fit a range of '
                                             'hyper-parameter values.'.format(name))
                row.append(float(v))
            hyper_rows = np.array([row], dtype=float).reshape(1, len(slots))
        hyper_rows = np.asarray(hyper_rows, dtype=float).reshape(-1, len(slots)) if len(slots) else np.zeros((max(1, len(hyper_rows)), 0))
        n = hyper_rows.shape[0]
        out = np.full((n, max(1, len(program))), np.nan)
        ts = self.formattedTimestamps if timestamps is None else timestamps
        done = set()
        for j, (kind, axis, model, k, seg, flg) in enumerate(program):
            if k is None:
                continue
            if isinstance(k, tuple):                       # ('shift', q) of a Deterministic model
                if id(model) in done:
                    continue
                done.add(id(model))
                cols = [i for i, sl in enumerate(slots) if sl[0] is model]
                # inside a serial model the sub-model counts its time from the break-/change-point that starts its segment
                # (reference transitionModels.py:770-776): that boundary's value of the SAME chain
                seg = [op[4] for op in program if op[0] == _abi.OP_DETERMINISTIC and op[2] is model][0]
                off_col = None
                if seg > 0:
                    bounds = [op for op in program if op[0] == _abi.OP_BREAKPOINT or (op[0] == _abi.OP_CHANGEPOINT and op[5] & 1)]
                    owner, kb = bounds[seg - 1][2], bounds[seg - 1][3]
                    off_col = [i for i, sl in enumerate(slots) if sl[0] is owner and sl[1] == kb][0]
                # (chains that share the model's hyper-parameters and its time offset share the table: a change-point study over
                #  two break-points evaluates each (parameters, first break-point) pair once, not once per chain)
                names_k = [model.hyperParameterNames[slots[i][1]] for i in cols]
                key_cols = cols + ([] if off_col is None else [off_col])
                if key_cols:
                    # rows of equal keys: per-column codes (1-D unique: a sort of n scalars each) combined into one integer code per chain
                    # -- np.unique(keys, axis=0) sorted n structured rows, a quarter of the host time of the reference's break-point study
                    code = np.zeros(n, dtype=np.int64)
                    for cidx in key_cols:
                        vals, inv_c = np.unique(hyper_rows[:, cidx], return_inverse=True)
                        code = code * len(vals) + np.asarray(inv_c).reshape(-1)
                    _, first, inv = np.unique(code, return_index=True, return_inverse=True)
                    uniq = hyper_rows[first][:, key_cols]
                else:
                    uniq, inv = np.zeros((1, 0)), np.zeros(n, dtype=np.intp)
                # (first segment of a serial model: the offset is 0 by definition (reference transitionModels.py:772), whatever a
                #  computeForwardPrior call through SerialTransitionModel._active left in the mutable model.tOffset; a stand-alone
                #  model (seg -1) reads its own attribute as in the reference)
                t_offs = uniq[:, -1] if off_col is not None else (np.zeros(len(uniq)) if seg == 0 else None)
                tables = model.shifts_many(names_k, uniq[:, :len(cols)], ts, resume_time, t_offsets=t_offs)
                # the model's 2 T columns in one gather + one scatter
                js = [jj for jj, op in enumerate(program) if isinstance(op[3], tuple) and op[2] is model]
                qs = [program[jj][3][1] for jj in js]
                inv = np.asarray(inv).reshape(-1)
                if js == list(range(js[0], js[0] + len(js))) and qs == list(range(tables.shape[1])):
                    out[:, js[0]:js[0] + len(js)] = tables[inv]           # (the usual layout: the 2 T columns follow their op in order)
                else:
                    out[:, js] = tables[inv][:, qs]
                continue
            col = [i for i, sl in enumerate(slots) if sl[0] is model and sl[1] == k][0]
            out[:, j] = hyper_rows[:, col]
        return out

    def _warnZero(self, phase):
        which = 'Forward pass distribution' if phase == 0 else 'Posterior distribution'
        if self.fitWarningCounter < 5:
            print('    ! WARNING: {} contains only zeros, check parameter boundaries!'.format(which))
            print('      Stopping inference process. Setting model evidence to zero.')
        elif self.fitWarningCounter == 5:
            print('    ! WARNING: Will omit further warnings about parameter boundaries.')
        self.fitWarningCounter += 1

    # ---- fit -----------------------------------------------------------------------------------------------------
    def fit(self, forwardOnly=False, evidenceOnly=False, silent=False):
        """
        Posterior sequence, posterior means, local evidence and log-evidence for the current hyper-parameter values
        (reference core.py:330-486), computed on the GPU.  ``posteriorSequence`` stays on the device until first read.
        """
        self._checkConsistency()
        if not silent:
            print('+ Started new fit:')
        self._formatData()
        if not silent:
            print('    + Formatted data.')
        if _tm_mod.needs_host_transition(self.transitionModel):
            return self._fitHostTransition(forwardOnly=forwardOnly, evidenceOnly=evidenceOnly, silent=silent)
        problem, program = self._compile(silent=silent)
        eng = _engine_mod.get_engine()
        T = len(self.formattedData)
        keep = not evidenceOnly
        if self._posterior_pending is not None:
            # the device buffer behind the previous fit's posterior is about to be reused.  A fit that stores a new sequence
            # replaces it anyway; an evidence-only fit leaves the previous posteriorSequence in place in the reference
            # (core.py:355-356 allocates only `if not evidenceOnly`), so it is brought to the host first.
            if keep:
                self._posterior_pending = None
            else:
                self._materialize_posterior()
        res = eng.fit(problem, self._opValueMatrix(program), forward_only=forwardOnly, evidence_only=evidenceOnly,
                      keep_posterior=keep, owner=self)
        self.lastTiming = res.timing
        self.logEvidence = float(res.log_evidence[0])
        self.localEvidence = res.local_evidence[0].copy()
        if res.abort_step[0] >= 0:
            # zero normaliser: the reference warns, sets logEvidence = -inf and returns early (core.py:390-400, 442-452)
            self._warnZero(int(res.abort_phase[0]))
            self.logEvidence = -np.inf
            if keep:
                eng.release_posterior(self)
                self._posterior_pending = None       # (the reference leaves np.empty garbage here; nothing is exposed)
                self._posteriorSequence = None
            return
        if not silent:
            print('    + Finished forward pass.')
            print('    + Log10-evidence: {:.5f}'.format(self.logEvidence / np.log(10)))
        if evidenceOnly:
            self.posteriorMeanValues = []
            return
        grid_size = list(self.gridSize)
        self._posteriorSequence = None
        self._posterior_pending = DevicePosterior(eng, 0, T, grid_size)
        self.posteriorMeanValues = res.posterior_mean[0].copy()
        if not silent:
            if not forwardOnly:
                print('    + Finished backward pass.')
            print('    + Computed mean parameter values.')

    # ---- the transition-model plug-in boundary: models the library has no program for ------------------------------------------
    _HOST_SLOT = 0x7fff0002            # carried-state slot of the library context used by the host-transition fit
    _host_transition_announced = frozenset()      # (model class names already announced on stderr; replaced, not mutated: `set` is a method here)

    def _fitHostTransition(self, forwardOnly=False, evidenceOnly=False, silent=False):
        """``Study.fit`` (reference core.py:330-486) for a transition model that is applied through the reference's plug-in
        interface, ``computeForwardPrior(posterior, t)`` / ``computeBackwardPrior(posterior, t)`` (transitionModels.py:49-63; called at
        core.py:411 and :467): a user-defined model, or a combination containing one.  Per time step the state makes a round trip
        over PCIe -- D2H, the model's own method on the host, H2D -- which is slow but exactly the reference's recursion; everything
        else of a step stays on the GPU as one-step problems through the same C-ABI entry point (``blhip_fit``): the likelihood
        product alpha = prior * L, its normaliser and evidence terms (:375-404), and backwards the posterior alpha_i * beta_i, its
        normalisation, sum(p / L) and the means (:436-464, :480-483) with the backward message handed over as
        ``blhip_problem.backward_init``, and beta * L (:467) as a one-step product.  Announced once per model class on stderr."""
        import sys
        tm, om = self.transitionModel, self.observationModel
        tm.study, tm.latticeConstant = self, self.latticeConstant
        key = type(tm).__name__
        if key not in Study._host_transition_announced:
            Study._host_transition_announced = Study._host_transition_announced | {key}
            sys.stderr.write('[bayesloop_amd] transition model "{}" has no device program: its computeForwardPrior / '
                             'computeBackwardPrior run on the host, one PCIe round trip of the state per time step (likelihood, '
                             'normalisation, evidence and means stay on the GPU)\n'.format(tm))
        gs = list(self.gridSize)
        if not 1 <= len(gs) <= 2:
            raise ConfigurationError('User-defined transition models are supported on grids with one or two parameters '
                                     '(got {}).'.format(len(gs)))
        eng = _engine_mod.get_engine()
        T = len(self.formattedData)
        ts = np.asarray(self.formattedTimestamps, dtype=float)
        dV = float(np.prod(self.latticeConstant))
        code = device_code(om)
        keep = not evidenceOnly
        full = keep and not forwardOnly
        if self._posterior_pending is not None:
            if keep:
                self._posterior_pending = None
            else:
                self._materialize_posterior()
        no_value = np.full((1, 1), np.nan)

        def step_problem(i, prior, backward_init=None):
            seg = self.formattedData[i]
            lik = None
            if code == _abi.OM_TABLE:
                lik = np.array([np.asarray(om.processedPdf(self.grid, seg), dtype=float) * np.ones(gs)])
            return FitProblem(obs_model=code, marginal=self.marginalGrid, lattice=self.latticeConstant,
                              data=np.asarray([seg], dtype=float), timestamps=ts[i:i + 1], prior=prior,
                              ops=[(_abi.OP_STATIC, 0, -1, 0)], lik=lik, seg_len=om.segmentLength,
                              carry_slot=Study._HOST_SLOT, backward_init=backward_init)

        def aborted(phase):
            self._warnZero(phase)
            self.logEvidence = -np.inf
            self._posterior_pending = None
            if keep:
                self._posteriorSequence = None

        prior = np.array(self._computePrior(silent=silent), dtype=float).reshape(gs)
        # full fits keep the distribution ENTERING each step (the backward pass multiplies it with the likelihood again on the
        # device), forward-only fits the filtered posteriors; the same (T, G) host array receives the posteriors
        seq = np.empty([T] + gs) if keep else None
        means = np.empty((len(gs), T)) if keep else []
        local = np.empty(T)
        logE = 0.0
        self.localEvidence = local
        for i in range(T):                                                            # core.py:372
            if forwardOnly:
                res = eng.fit(step_problem(i, prior), no_value, forward_only=True, keep_posterior=True, owner=None)
            else:
                res = eng.fit(step_problem(i, prior), no_value, evidence_only=True, carry=True)
            if res.abort_step[0] >= 0:                                               # :390-400
                return aborted(0)
            local[i] = res.local_evidence[0, 0]                                      # :404  norm * dV
            logE += np.log(local[i] / dV)                                            # :403
            if forwardOnly:
                alpha = np.array(eng.posterior(0, 1, gs)[0])
                seq[i] = alpha                                                       # :408
                means[:, i] = res.posterior_mean[0, :, 0]
            else:
                alpha = eng.carry_read(Study._HOST_SLOT, 0, gs)
                if full:
                    seq[i] = prior
            prior = np.asarray(tm.computeForwardPrior(alpha, ts[i]), dtype=float).reshape(gs)      # :411
        logE += np.log(dV)                                                           # :417
        self.logEvidence = float(logE)
        self.lastTiming = eng.last_timing()
        if not silent:
            print('    + Finished forward pass.')
            print('    + Log10-evidence: {:.5f}'.format(self.logEvidence / np.log(10)))
        if evidenceOnly:
            self.posteriorMeanValues = []
            return
        if full:
            beta = np.ones(gs) / float(np.prod(gs))                                  # :424-425
            for i in range(T - 1, -1, -1):                                           # :434
                res = eng.fit(step_problem(i, seq[i], backward_init=beta), no_value, keep_posterior=True, owner=None)
                if res.abort_step[0] >= 0:                                           # :442-452
                    return aborted(1)
                local[i] = res.local_evidence[0, 0]                                  # :463-464
                means[:, i] = res.posterior_mean[0, :, 0]                            # :480-483
                seq[i] = eng.posterior(0, 1, gs)[0]                                  # :436-441
                if i == 0:
                    break
                r2 = eng.fit(step_problem(i, beta), no_value, evidence_only=True, carry=True)       # beta * L, :467
                with np.errstate(all='ignore'):
                    if r2.abort_step[0] >= 0:
                        c = np.zeros(gs)
                    else:
                        c = eng.carry_read(Study._HOST_SLOT, 0, gs) * (r2.local_evidence[0, 0] / dV)
                    beta = np.asarray(tm.computeBackwardPrior(c, ts[i]), dtype=float).reshape(gs)
                    beta = beta / np.sum(beta)                                       # :470
            if not silent:
                print('    + Finished backward pass.')
        eng.release_posterior(None)
        self._posterior_pending = None
        self._posteriorSequence = seq
        self.posteriorMeanValues = means
        if not silent:
            print('    + Computed mean parameter values.')

    def optimize(self, parameterList=[], forwardOnly=False, **kwargs):
        """COBYLA maximisation of the log-evidence over hyper-parameters (reference core.py:488-565); every
        objective evaluation is an evidence-only fit on the GPU."""
        from scipy.optimize import minimize
        self.selectedHyperParameters = [parameterList] if isinstance(parameterList, str) else list(parameterList)
        print('+ Starting optimization...')
        self._checkConsistency()
        if not self.selectedHyperParameters:
            points = self._unpackChangepointNames() + self._unpackBreakpointNames()
            self.selectedHyperParameters = [n for n in self._unpackAllHyperParameters(values=False) if n not in points]
        x0 = self._unpackSelectedHyperParameters()
        if len(x0) == 0:
            self.selectedHyperParameters = []
            raise ConfigurationError('No parameters to optimize. Check parameter names.')

        def objective(x):
            self._setSelectedHyperParameters(x)
            self.fit(evidenceOnly=True, silent=True)
            print('    + Log10-evidence: {:.5f}'.format(self.logEvidence / np.log(10)), '- Parameter values:', x)
            return -self.logEvidence

        result = minimize(objective, x0, method='COBYLA', **kwargs)
        print('+ Finished optimization.')
        self._setSelectedHyperParameters(result.x)
        self.fit(forwardOnly=forwardOnly)
        self.selectedHyperParameters = []

    # ---- accessors (host-side consumers of the fit results) -------------------------------------------------------
    def _parameterIndex(self, name):
        names = list(self.observationModel.parameterNames)
        if name not in names:
            raise PostProcessingError('Wrong parameter name. Available options: {0}'.format(names))
        return names.index(name)

    def _requirePosterior(self):
        post = self.posteriorSequence
        if post is None or len(post) == 0:
            raise PostProcessingError('Cannot plot posterior sequence as it has not yet been computed. Run complete fit.')
        return post

    def getParameterMeanValues(self, name):
        return self.posteriorMeanValues[self._parameterIndex(name)]

    @_plots('param_dist')
    def getParameterDistribution(self, t, name, plot=False, density=True, **kwargs):
        """Marginal distribution of one parameter at time stamp ``t`` or time-averaged (``t='avg'``).  While the
        posterior sequence is still on the GPU the reduction happens there (no (T, G) copy)."""
        k = self._parameterIndex(name)
        pending = self._posterior_pending
        if isinstance(t, str) and t == 'avg':
            if pending is not None and hasattr(pending, 'time_average'):
                dist = pending.time_average()
            else:
                post = self._requirePosterior()
                dist = np.sum(post, axis=0) / len(post)
        else:
            if t not in self.formattedTimestamps:
                raise PostProcessingError('Supplied time ({}) does not exist in data or is out of range.'.format(t))
            index = list(self.formattedTimestamps).index(t)
            if pending is not None and hasattr(pending, 'marginal'):
                marginal = pending.marginal(k)[index]
                if density:
                    marginal = marginal / self.latticeConstant[k]
                return self.marginalGrid[k], marginal
            dist = self._requirePosterior()[index]
        axes = tuple(a for a in range(len(self.gridSize)) if a != k)
        marginal = np.sum(dist, axis=axes) if axes else np.array(dist)
        if density:
            marginal = marginal / self.latticeConstant[k]
        return self.marginalGrid[k], marginal

    def getPD(self, t, name, plot=False, density=True, **kwargs):
        return self.getParameterDistribution(t, name, plot=plot, density=density, **kwargs)

    def simulate(self, x, t=None, density=False):
        """Probability (density) of the observation values ``x`` under the inferred parameter distribution of time stamp
        ``t`` (or its time average), reference core.py:566-597.  Only the (time-averaged) distribution of one step leaves
        the GPU: G doubles, not the (T, G) sequence."""
        om = self.observationModel
        if om.segmentLength > 1:
            raise NotImplementedError('Method "simulate" is only available for observation models with segment length 1.')
        pending = self._posterior_pending
        if t is None:
            if pending is not None and hasattr(pending, 'time_average'):
                post = pending.time_average()
            else:
                seq = self._requirePosterior()
                post = np.sum(seq, axis=0) / len(seq)
        else:
            if t not in self.formattedTimestamps:
                raise PostProcessingError('Supplied time ({}) does not exist in data or is out of range.'.format(t))
            index = list(self.formattedTimestamps).index(t)
            post = pending.row(index) if pending is not None and hasattr(pending, 'row') else self._requirePosterior()[index]
        prob = np.array([np.sum(om.pdf(self.grid, [xi]) * post) for xi in x])
        if not density:
            prob /= np.sum(prob)
        return prob

    @_plots('param_dists')
    def getParameterDistributions(self, name, plot=False, density=True, **kwargs):
        """Time series of marginal posterior distributions of one parameter: (values, (T, n) array); reduced on the GPU
        while the posterior sequence is still there."""
        k = self._parameterIndex(name)
        pending = self._posterior_pending
        if pending is not None and hasattr(pending, 'marginal'):
            marginal = pending.marginal(k)
        else:
            post = self._requirePosterior()
            axes = tuple(a + 1 for a in range(len(self.gridSize)) if a != k)
            marginal = np.sum(post, axis=axes) if axes else np.array(post)
        if density:
            marginal = marginal / self.latticeConstant[k]
        return self.marginalGrid[k], marginal

    def getPDs(self, name, plot=False, density=True, **kwargs):
        return self.getParameterDistributions(name, plot=plot, density=density, **kwargs)

    def plotParameterEvolution(self, name, color='b', gamma=0.5, **kwargs):
        """Image of the marginal posterior of one parameter over time (gamma-corrected) with the posterior means on top
        (reference core.py:1007-1069)."""
        from . import plotting
        k = self._parameterIndex(name)
        x, p = self.getParameterDistributions(name, density=False)
        plotting.evolution(self.formattedTimestamps, self.boundaries[k], p, self.getParameterMeanValues(name), name,
                           color=color, gamma=gamma, **kwargs)

    def plot(self, name, **kwargs):
        """Evolution of a parameter, or its distribution at time stamp ``t=...`` (reference core.py:1071-1096)."""
        density = kwargs.pop('density', True)
        if 't' in kwargs:
            self.getParameterDistribution(kwargs.pop('t'), name, plot=True, density=density, **kwargs)
        else:
            self.plotParameterEvolution(name, color=kwargs.pop('color', 'b'), gamma=kwargs.pop('gamma', 0.5), **kwargs)


class HyperStudy(Study):
    """Hyper-parameter inference over a grid of hyper-parameter values (reference core.py:1118-1495).

    All hyper-grid points are independent forward-backward chains; they run batched on the GPU (one kernel launch per
    time step for the whole batch) and, when a communicator is attached (:mod:`bayesloop_amd.dist`), sharded over the
    GPUs of a node round-robin (the reference's ``_parallelFit`` hands out ``np.array_split`` chunks; same results).
    """

    def __init__(self, silent=False):
        super(HyperStudy, self).__init__(silent=silent)
        self.hyperGrid = []
        self.hyperGridValues = []
        self.hyperGridConstant = []
        self.flatHyperParameters = []
        self.flatHyperParameterNames = []
        self.flatHyperPriors = []
        self.flatHyperPriorValues = []
        self.hyperParameterDistribution = None
        self.averagePosteriorSequence = None
        self.logEvidenceList = []
        self.localEvidenceList = []
        self.communicator = None        # bayesloop_amd.dist communicator; None = single GPU
        if not silent:
            print('  --> Hyper-study')

    def _unpackAllHyperPriors(self):
        priors = []
        for m, k, name in self._hyperSlots():
            prior = getattr(m, 'prior', None)
            per_parameter = isinstance(m, (SerialTransitionModel, BivariateRandomWalk, AlphaStableRandomWalk, Deterministic))
            if not per_parameter and type(m).__module__ != _tm_mod.__name__ and isinstance(prior, (list, tuple)):
                # a user-defined model with several hyper-parameters lists one prior per hyper-parameter; the reference flattens the
                # nested prior list (core.py:1501-1533)
                per_parameter = len(prior) == len(m.hyperParameterNames) > 1
            priors.append(prior[k] if per_parameter else prior)
        return priors

    def _createHyperGrid(self, silent=False):
        """Hyper-grid, its lattice constants and the joint hyper-prior (reference core.py:1142-1245)."""
        self.flatHyperParameters = self._unpackAllHyperParameters()
        self.flatHyperParameterNames = self._unpackAllHyperParameters(values=False)
        self.flatHyperPriors = self._unpackAllHyperPriors()
        for k, v in enumerate(self.flatHyperParameters):
            if isinstance(v, str) and v == 'all':
                self.flatHyperParameters[k] = self.formattedTimestamps[:-1]

        if len(self.flatHyperParameterNames) > 0:
            mesh = np.meshgrid(*self.flatHyperParameters, indexing='ij')
            self.hyperGridValues = np.array([m.ravel() for m in mesh]).T
        else:
            self.hyperGridValues = np.array([])

        constants = []
        for values in self.flatHyperParameters:
            c = 1
            if isinstance(values, Iterable) and len(values) > 1:
                a = np.array(values)
                d = a[1:] - a[:-1]
                if np.all(np.abs(d[1:] - d[:-1]) < 10 ** -10):
                    c = np.abs(d[0])
            constants.append(c)
        self.hyperGridConstant = np.array(constants)

        priorValuesList, priorNames = [], []
        for prior, values, c, name in zip(self.flatHyperPriors, self.flatHyperParameters, self.hyperGridConstant,
                                          self.flatHyperParameterNames):
            if prior is None:
                pv = np.ones_like(values, dtype=float)
                pv /= np.sum(pv)
                pv /= c
                priorNames.append('uniform')
            elif hasattr(prior, '__call__'):
                try:
                    pv = np.array([prior(v) for v in np.atleast_1d(values)], dtype=float)
                    pv = pv / np.sum(pv) / c
                except Exception:
                    raise ConfigurationError('Failed to set hyper-prior for "{}" from function "{}".'
                                             .format(name, getattr(prior, '__name__', 'callable')))
                priorNames.append(getattr(prior, '__name__', 'callable'))
            elif isinstance(prior, Iterable):
                if len(prior) != len(values):
                    raise ConfigurationError('Failed to set hyper-prior for "{}" from list/array.'.format(name))
                pv = np.array(prior, dtype=float)
                pv = pv / np.sum(pv) / c
                priorNames.append('list/array')
            else:
                pv = self._sympyHyperPrior(prior, values, name)
                priorNames.append('sympy')
            priorValuesList.append(pv)

        if len(self.flatHyperParameterNames) > 0:
            mesh = np.meshgrid(*priorValuesList, indexing='ij')
            self.flatHyperPriorValues = np.prod(np.array([m.ravel() for m in mesh]).T, axis=1)
            if not silent and len(self.hyperGridValues) > 1:
                print('+ Set hyper-prior(s): {}'.format(priorNames))
        else:
            self.flatHyperPriorValues = np.array([1])

    @staticmethod
    def _sympyHyperPrior(prior, values, name):
        try:
            from sympy import lambdify, abc
            from sympy.stats import density
            pdf = lambdify([abc.x], density(prior)(abc.x), modules=['numpy'])
            return np.asarray(pdf(np.asarray(values, dtype=float)), dtype=float)
        except Exception:
            raise ConfigurationError('Failed to set hyper-prior for "{}".'.format(name))

    def fit(self, forwardOnly=False, evidenceOnly=False, silent=False, nJobs=1, customHyperGrid=False):
        """
        Fits every hyper-grid point and averages the models with their evidence (reference core.py:1247-1441).

        ``nJobs`` (reference core.py:1307-1340: a pool of worker processes over chunks of the hyper-grid): with ``nJobs > 1`` and
        more than one GPU visible to this process the hyper-grid points are dealt out to min(nJobs, GPUs) devices, one host
        thread and one library context per device, and the results are merged over xGMI (``bayesloop_amd.dist.LocalGroup``) --
        no launcher, nothing else to set up.  On ONE GPU all chains are batched on that device whatever ``nJobs`` says.  Under a
        one-process-per-GPU launcher set ``self.communicator`` (``bl.dist.RcclCommunicator``) instead.
        """
        self.fitWarningCounter = 0
        self._formatData()
        if not customHyperGrid:
            self._createHyperGrid(silent=silent)
            names = self._unpackChangepointNames() + self._unpackBreakpointNames()
            if len(names) > 1:
                cols = [self.flatHyperParameterNames.index(n) for n in names]
                for v in self.hyperGridValues[:, cols]:
                    if np.unique(v).size < v.size:
                        raise ConfigurationError('Detected multiple change-/break-points with identical values and/or '
                                                 'overlapping value intervals. Use "ChangepointStudy" instead of '
                                                 '"HyperStudy" for such cases.')
        self._checkConsistency()
        self.logEvidenceList, self.localEvidenceList = [], []
        if len(self.hyperGridValues) <= 1:
            if not silent:
                print('+ At most one combination of hyper-parameter values, switching to standard fit method.')
            if len(self.hyperGridValues) == 1:
                self._setAllHyperParameters(self.hyperGridValues[0])
            Study.fit(self, forwardOnly=forwardOnly, evidenceOnly=evidenceOnly, silent=silent)
            if len(self.hyperGridValues) == 1:
                self._setAllHyperParameters(self.flatHyperParameters)
            return

        if not silent:
            print('+ Started new fit.')
            print('    + {} analyses to run.'.format(len(self.hyperGridValues)))
        if _tm_mod.needs_host_transition(self.transitionModel):
            return self._fitHostTransitionHyper(forwardOnly, evidenceOnly, silent)
        # one representative value per hyper-parameter while compiling (values come from the hyper-grid rows)
        self._setAllHyperParameters(self.hyperGridValues[0])
        try:
            problem, program = self._compile(silent=True)
        finally:
            self._setAllHyperParameters(self.flatHyperParameters)
        op_values = self._opValueMatrix(program, np.asarray(self.hyperGridValues, dtype=float))
        prior_values = np.asarray(self.flatHyperPriorValues, dtype=float)

        from . import dist as _dist
        root_engine = _engine_mod.get_engine()
        devices = [getattr(root_engine, 'device', 0)]
        if self.communicator is None and nJobs and int(nJobs) > 1 and hasattr(root_engine, 'ctx'):
            devices = _dist.local_devices(int(nJobs), getattr(root_engine, 'device', 0))
        out = None
        if len(devices) > 1 and not _dist.multi_gpu_disabled():
            # Several GPUs from this one process.  The exchange is checked end to end on every fit (the merged accumulator's per-step
            # sums against the parts', dist.sharded_hyper_fit); if anything on that path fails -- a peer copy, a context on another
            # device, the checksum -- the fit is repeated on the root device alone and the process stops using the path: the answer
            # never depends on the multi-device plumbing.  BLHIP_NJOBS_MULTI_GPU=0 switches the path off, =strict re-raises.
            try:
                engines, seen = [], set()
                for dev in devices:        # (a repeated ordinal -- BLHIP_NJOBS_DEVICES=0,0, tests -- gets a context of its own)
                    engines.append(_engine_mod.engine_for_device(dev) if dev not in seen else _engine_mod.extra_engine(dev))
                    seen.add(dev)
                out = _dist.local_sharded_hyper_fit(engines, problem, op_values, prior_values, forward_only=forwardOnly,
                                                    evidence_only=evidenceOnly, owner=self)
            except Exception as exc:       # noqa: BLE001 -- anything the multi-device path raises; the single-device path is the reference
                if _dist.multi_gpu_strict():
                    raise
                _dist.disable_multi_gpu('%s: %s' % (type(exc).__name__, exc))
                out = None
        if out is None:
            out = _dist.sharded_hyper_fit(root_engine, problem, op_values, prior_values, self.communicator,
                                          forward_only=forwardOnly, evidence_only=evidenceOnly, owner=self)
        self.lastTiming = out['timing']
        self.lastTimingPerDevice = out.get('per_rank_timing')
        self.logEvidenceList = list(out['log_evidence'])
        localList = out['local_evidence']
        n_abort = int(np.sum(out['abort_step'] >= 0))
        for _ in range(min(n_abort, 6)):
            self._warnZero(0)
        self.fitWarningCounter += max(0, n_abort - 6)

        if not evidenceOnly:
            self.averagePosteriorSequence = None
            self._posteriorSequence = None
            self._posterior_pending = out['posterior']          # callable or None (non-root ranks)
            if not silent:
                print('    + Computed average posterior sequence')

        # hyper-parameter distribution and evidence of the average model (reference core.py:1391-1410)
        with np.errstate(divide='ignore'):
            logHPD = np.asarray(out['log_evidence'], dtype=float) + np.log(prior_values) + np.sum(np.log(self.hyperGridConstant))
        scaled = logHPD - np.amax(logHPD)
        self.hyperParameterDistribution = np.exp(scaled)
        self.hyperParameterDistribution /= np.sum(self.hyperParameterDistribution)
        self.hyperParameterDistribution /= np.prod(self.hyperGridConstant)
        self.logEvidence = float(_logsumexp(logHPD))                            # :1405 scipy.special.logsumexp
        if not silent:
            print('    + Computed hyper-parameter distribution')
            print('    + Log10-evidence of average model: {:.5f}'.format(self.logEvidence / np.log(10)))
        self.localEvidence = np.sum((np.asarray(localList).T * prior_values).T, axis=0)
        if not evidenceOnly:
            self.posteriorMeanValues = out['posterior_mean']
        self.localEvidenceList = []
        self._setAllHyperParameters(self.flatHyperParameters)
        if not silent:
            print('+ Finished fit.')

    def _fitHostTransitionHyper(self, forwardOnly, evidenceOnly, silent):
        """A hyper-study over a transition model that is applied on the host (Study._fitHostTransition): one fit per hyper-grid point
        (reference core.py:1349-1361); unless ``evidenceOnly`` every finite chain's posterior sequence -- which the per-step round trips
        of that fit have left on the host -- is folded into the device accumulator with its weight (:1362-1366,
        ``blhip_accum_fold_host``), normalised and reduced there (:1375-1382, :1416-1419); then the hyper-parameter distribution and
        the evidence of the average model (:1391-1410)."""
        prior_values = np.asarray(self.flatHyperPriorValues, dtype=float)
        self.logEvidenceList, localList = [], []
        eng = _engine_mod.get_engine()
        want_post = not evidenceOnly
        n_fold, begun = 0, False
        try:
            for k, row in enumerate(self.hyperGridValues):
                self._setAllHyperParameters(row)
                Study.fit(self, forwardOnly=forwardOnly, evidenceOnly=evidenceOnly, silent=True)
                self.logEvidenceList.append(self.logEvidence)
                localList.append(np.array(self.localEvidence))
                if want_post and np.isfinite(self.logEvidence) and self._posteriorSequence is not None:
                    seq = np.asarray(self._posteriorSequence, dtype=float)
                    if not begun:
                        eng.accum_begin(seq.shape[0], int(np.prod(seq.shape[1:])), owner=self)
                        begun = True
                    with np.errstate(divide='ignore'):
                        eng.accum_fold_host(seq, self.logEvidence + np.log(prior_values[k]))
                    n_fold += 1 if prior_values[k] > 0 else 0
        finally:
            self._setAllHyperParameters(self.flatHyperParameters)
        if want_post:
            self._posteriorSequence = None
            self._posterior_pending = None
            self.averagePosteriorSequence = None
            self.posteriorMeanValues = []
            if n_fold > 0:
                # (marginal grids and lattice are all accum_finalize reads of the problem: per-step normalisation and the means of the average)
                T = len(self.formattedData)
                shell = FitProblem(obs_model=device_code(self.observationModel), marginal=self.marginalGrid, lattice=self.latticeConstant,
                                   data=np.zeros((T, 1)), timestamps=np.zeros(T), prior=np.ones(int(np.prod(self.gridSize))), ops=[])
                self.posteriorMeanValues = eng.accum_finalize(shell)
                self._posterior_pending = DevicePosterior(eng, 1, T, list(self.gridSize))
                if hasattr(eng, 'accum_set_owner'):          # (the per-point fits released the engine's results: the average is this study's)
                    eng.accum_set_owner(self)
                if not silent:
                    print('    + Computed average posterior sequence')
        with np.errstate(divide='ignore'):
            logHPD = np.array(self.logEvidenceList) + np.log(prior_values) + np.sum(np.log(self.hyperGridConstant))
        scaled = logHPD - np.amax(logHPD)
        self.hyperParameterDistribution = np.exp(scaled)
        self.hyperParameterDistribution /= np.sum(self.hyperParameterDistribution)
        self.hyperParameterDistribution /= np.prod(self.hyperGridConstant)
        self.logEvidence = float(_logsumexp(logHPD))
        self.localEvidence = np.sum((np.array(localList).T * prior_values).T, axis=0)
        self.localEvidenceList = []
        if not silent:
            print('    + Computed hyper-parameter distribution')
            print('    + Log10-evidence of average model: {:.5f}'.format(self.logEvidence / np.log(10)))
            print('+ Finished fit.')

    @property
    def averagePosteriorSequence(self):
        return self.posteriorSequence if self._avg_is_posterior else self._averagePosteriorSequence

    @averagePosteriorSequence.setter
    def averagePosteriorSequence(self, value):
        self._avg_is_posterior = value is None
        self._averagePosteriorSequence = value

    def optimize(self, *args, **kwargs):
        raise NotImplementedError('HyperStudy object has no optimizing method.')

    # ---- accessors -------------------------------------------------------------------------------------------------
    def _hyperGridSteps(self):
        return [len(x) if isinstance(x, Iterable) and not isinstance(x, str) else 1 for x in self.flatHyperParameters]

    @_plots('hyper_dist')
    def getHyperParameterDistribution(self, name, plot=False, **kwargs):
        """Marginal distribution of one hyper-parameter: (values, probabilities) (reference core.py:1535-1585)."""
        if len(self.hyperGridValues) < 2:
            raise PostProcessingError('At least two combinations of hyper-parameter values need to be fitted to '
                                      'evaluate a hyper-parameter distribution. Check transition model.')
        k = self._getHyperParameterIndex(self.transitionModel, name)
        dist = np.asarray(self.hyperParameterDistribution).reshape(self._hyperGridSteps(), order='C')
        axes = tuple(a for a in range(dist.ndim) if a != k)
        marginal = np.sum(dist, axis=axes) if axes else np.array(dist)
        marginal = marginal * np.prod(self.hyperGridConstant)
        return self.flatHyperParameters[k], marginal

    def getHPD(self, name, plot=False, **kwargs):
        return self.getHyperParameterDistribution(name, plot=plot, **kwargs)

    @_plots('joint')
    def getJointHyperParameterDistribution(self, names, plot=False, figure=None, subplot=111, **kwargs):
        """Joint distribution of two hyper-parameters: (x, y, probabilities) (reference core.py:1593-1694)."""
        if len(self.hyperGridValues) < 2:
            raise PostProcessingError('At least two combinations of hyper-parameter values need to be fitted to '
                                      'evaluate a hyper-parameter distribution. Check transition model.')
        if not isinstance(names, Iterable) or isinstance(names, str) or len(names) != 2:
            raise PostProcessingError('A list of exactly two hyper-parameters has to be provided.')
        idx = [self._getHyperParameterIndex(self.transitionModel, n) for n in names]
        switch = idx[0] > idx[1]
        lo, hi = sorted(idx)
        dist = np.asarray(self.hyperParameterDistribution).reshape(self._hyperGridSteps(), order='C')
        axes = tuple(a for a in range(dist.ndim) if a not in (lo, hi))
        marginal = np.sum(dist, axis=axes) if axes else np.array(dist)
        marginal = marginal * np.prod(self.hyperGridConstant)
        x, y = self.flatHyperParameters[lo], self.flatHyperParameters[hi]
        if switch:
            x, y, marginal = y, x, marginal.T
        return x, y, marginal

    def getJHPD(self, names, plot=False, figure=None, subplot=111, **kwargs):
        return self.getJointHyperParameterDistribution(names, plot=plot, figure=figure, subplot=subplot, **kwargs)

    def plot(self, name, **kwargs):
        """Evolution of a parameter, its distribution at ``t=...``, or the distribution of a hyper-parameter
        (reference core.py:1702-1741)."""
        density = kwargs.pop('density', True)
        if 't' in kwargs:
            self.getParameterDistribution(kwargs.pop('t'), name, plot=True, density=density, **kwargs)
        elif name in self._unpackAllHyperParameters(values=False):
            self.getHyperParameterDistribution(name, plot=True, **kwargs)
        else:
            self.plotParameterEvolution(name, color=kwargs.pop('color', 'b'), gamma=kwargs.pop('gamma', 0.5), **kwargs)


class ChangepointStudy(HyperStudy):
    """Change-point inference: a HyperStudy over ordered combinations of change-point times
    (reference core.py:1743-1852)."""

    def __init__(self, silent=False):
        super(ChangepointStudy, self).__init__(silent=silent)
        self.allHyperGridValues = []
        self.allHyperPriorValues = []
        self.mask = []
        if not silent:
            print('  --> Change-point analysis')

    def fit(self, forwardOnly=False, evidenceOnly=False, silent=False, nJobs=1):
        self._formatData()
        serials = [m for m, k, name in self._hyperSlots() if isinstance(m, SerialTransitionModel)]
        if len(set(id(m) for m in serials)) > 1:
            raise NotImplementedError('Multiple instances of SerialTransition models are currently not supported by '
                                      'ChangepointStudy.')
        changepoints = self._unpackChangepointNames()
        breakpoints = self._unpackBreakpointNames()
        if len(changepoints) > 0 and len(breakpoints) > 0:
            raise NotImplementedError('Detected both change-points (Changepoint transition model) and break-points '
                                      '(SerialTransitionModel). Currently, only one type is supported in a single '
                                      'transition model.')
        if len(changepoints) == 0 and len(breakpoints) == 0:
            raise ConfigurationError('No change-points or break-points detected in transition model. Check transition '
                                     'model.')
        if len(changepoints) == 0:
            changepoints = breakpoints
        self.flatHyperParameters = self._unpackAllHyperParameters()
        self.flatHyperParameterNames = self._unpackAllHyperParameters(values=False)
        if not silent:
            print('+ Detected {} change-point(s) in transition model: {}'.format(len(changepoints), changepoints))

        self._createHyperGrid(silent=silent)
        self.allHyperGridValues = self.hyperGridValues[:]
        self.allHyperPriorValues = self.flatHyperPriorValues[:]

        # keep ordered combinations of change-point times only and re-weight the prior (reference core.py:1823-1834)
        cols = [self.flatHyperParameterNames.index(n) for n in changepoints]
        points = self.allHyperGridValues[:, cols]
        self.mask = np.ones(len(points), dtype=bool)
        for a in range(points.shape[1] - 1):
            self.mask &= points[:, a] < points[:, a + 1]
        self.hyperGridValues = self.allHyperGridValues[self.mask]
        self.flatHyperPriorValues = self.allHyperPriorValues[self.mask] * \
            (np.sum(self.allHyperPriorValues) / np.sum(self.allHyperPriorValues[self.mask]))

        HyperStudy.fit(self, forwardOnly=forwardOnly, evidenceOnly=evidenceOnly, silent=silent, nJobs=nJobs,
                       customHyperGrid=True)

        # scatter back to the full grid, invalid combinations get probability zero (reference core.py:1846-1852)
        if self.hyperParameterDistribution is not None and len(self.hyperGridValues) > 1:
            full = np.zeros(len(self.allHyperGridValues))
            full[self.mask] = self.hyperParameterDistribution
            self.hyperParameterDistribution = full
        full = np.zeros(len(self.allHyperPriorValues))
        full[self.mask] = self.flatHyperPriorValues
        self.flatHyperPriorValues = full

    @_plots('duration')
    def getDurationDistribution(self, names, plot=False, **kwargs):
        """Distribution of the number of time steps between two change-points (reference core.py:1875-1924)."""
        if not isinstance(names, Iterable) or isinstance(names, str) or len(names) != 2:
            raise PostProcessingError('A list of exactly two hyper-parameters has to be provided.')
        idx = sorted(self._getHyperParameterIndex(self.transitionModel, n) for n in names)
        values = self.hyperGridValues[:, idx].T
        duration = np.unique(values[1] - values[0])
        dist = np.zeros(len(duration))
        for k, v in enumerate(self.allHyperGridValues[:, idx]):
            if v[1] > v[0]:
                j = np.where(duration.round(10) == (v[1] - v[0]).round(10))[0][0]
                dist[j] += self.hyperParameterDistribution[k]
        return duration, dist / np.sum(dist)

    def getDD(self, names, plot=False, **kwargs):
        return self.getDurationDistribution(names, plot=plot, **kwargs)


class OnlineStudy(HyperStudy):
    """Model selection on a data stream (reference core.py:1963-2226): ``step(dataPoint)`` advances the forward filter of
    EVERY (transition model, hyper-parameter value) pair by one data point and re-weights them with their evidence.

    On the GPU each pair is a chain whose filtered distribution stays on the device between calls ("carried state",
    include/blhip.h BLHIP_CARRY / BLHIP_RESUME): a step is one fused forward-step launch per transition model over all of
    its hyper-parameter values, and the evidence-weighted mixtures the study exposes (``marginalizedPosterior``,
    ``transitionModelPosterior``) are reduced on the device.  The host keeps the reference's evidence bookkeeping
    (core.py:2169-2214) on the handful of per-chain scalars.
    """
    _slot_counter = [0]

    def __init__(self, storeHistory=False, silent=False):
        super(OnlineStudy, self).__init__(silent=silent)
        self.firstStep = True
        self.storeHistory = storeHistory
        # the competing transition models and what the hyper-grid machinery produced for each of them (lists, one entry per model)
        for per_model in ('transitionModels', 'transitionModelNames', 'tmCounts', 'hyperParameterValues', 'allFlatHyperParameterValues',
                          'hyperParameterNames', 'hyperGridConstants', 'hyperPrior', 'hyperPriorValues'):
            setattr(self, per_model, [])
        # what a step updates (None until the first one) ...
        for updated in ('tmCount', 'logEvidenceList', 'hyperLogEvidenceList', 'transitionModelPrior', 'marginalizedPosterior',
                        'hyperParameterDistribution', 'transitionModelDistribution', 'localTransitionModelDistribution'):
            setattr(self, updated, None)
        # ... and the histories it appends to (storeHistory)
        self.posteriorMeanValues = []
        self._posteriorSequence = []
        for history in ('hyperParameterSequence', 'transitionModelSequence', 'localTransitionModelSequence'):
            setattr(self, history, [])
        self._slots = []            # carry slot of every transition model in the engine's context
        self._device = []           # per transition model: (ops, op_values, reset prior, indep prior)
        if not silent:
            print('  --> Online study')

    # the history is a plain list of host arrays (the reference appends a copy per step, core.py:2219)
    @property
    def posteriorSequence(self):
        return self._posteriorSequence

    @posteriorSequence.setter
    def posteriorSequence(self, value):
        self._posteriorSequence = value

    def __del__(self):
        try:
            eng = _engine_mod.get_engine()
            for s in self._slots:          # (every instance allocates its own slots: a copy never shares them)
                eng.carry_release(s)
        except Exception:
            pass

    # The carried per-chain filter states live in the engine's context, not in the object: a pickled copy takes them along
    # as host arrays and re-creates them in slots of its own, so that `bl.save` / `bl.load` / `copy.deepcopy` of an
    # OnlineStudy can go on stepping (reference fileIO.py:10-37 pickles the whole study, parameterPosterior included).
    def __getstate__(self):
        state = Study.__getstate__(self)
        carried = None
        if not self.firstStep and self._slots:
            eng = _engine_mod.get_engine()
            carried = [np.array([eng.carry_read(s, j, self.gridSize) for j in range(c)])
                       for s, c in zip(self._slots, self.tmCounts)]
        state['_carried'] = carried
        state['_slots'] = []
        return state

    def __setstate__(self, state):
        carried = state.pop('_carried', None)
        self.__dict__.update(state)
        self._slots = []
        if carried is not None:
            eng = _engine_mod.get_engine()
            for st in carried:
                OnlineStudy._slot_counter[0] += 1
                self._slots.append(OnlineStudy._slot_counter[0])
                eng.carry_write(self._slots[-1], st)

    # ---- configuration (reference core.py:2007-2060) -------------------------------------------------------------------
    def addTransitionModel(self, name, transitionModel):
        """One more competing transition model (reference core.py:1986-2026): its hyper-grid is built once, by the HyperStudy machinery,
        and filed per model; a model without hyper-parameters counts as one chain."""
        self.setTransitionModel(transitionModel, silent=True)
        self._createHyperGrid(silent=True)
        filed = ((self.transitionModels, transitionModel), (self.transitionModelNames, name),
                 (self.hyperParameterValues, self.hyperGridValues[:]), (self.allFlatHyperParameterValues, self.flatHyperParameters),
                 (self.hyperParameterNames, self.flatHyperParameterNames[:]), (self.hyperGridConstants, self.hyperGridConstant[:]),
                 (self.hyperPrior, self.flatHyperPriors[:]), (self.hyperPriorValues, self.flatHyperPriorValues[:]))
        for per_model, entry in filed:
            per_model.append(entry)
        self.tmCounts = [max(len(values), 1) for values in self.hyperParameterValues]
        self.tmCount = int(np.sum(self.tmCounts))
        n_comb = len(self.hyperGridValues)
        print('+ Added transition model: {} ({} combination(s) of the following hyper-parameters: {})'
              .format(name, n_comb, self.hyperParameterNames[-1]) if n_comb > 0
              else '+ Added transition model: {} (no hyper-parameters)'.format(name))

    def addTM(self, name, transitionModel):
        self.addTransitionModel(name, transitionModel)

    def add(self, name, transitionModel):
        self.addTransitionModel(name, transitionModel)

    def setTransitionModelPrior(self, transitionModelPrior, silent=False):
        if not (isinstance(transitionModelPrior, Iterable) and len(transitionModelPrior) == len(self.transitionModels)):
            raise ConfigurationError('Length of transition model prior ({}) does not fit number of transition models '
                                     '({})'.format(len(transitionModelPrior), len(self.transitionModels)))
        self.transitionModelPrior = np.array(transitionModelPrior, dtype=float)
        if not np.sum(transitionModelPrior) == 1.:
            print('+ WARNING: Transition model prior does not sum up to one. Will re-normalize.')
            self.transitionModelPrior /= np.sum(self.transitionModelPrior)
        if not silent:
            print('+ Set custom transition model prior.')

    def fit(self, *args, **kwargs):
        raise NotImplementedError('OnlineStudy object has no "fit" method. Use "step" instead.')

    # ---- one data point (reference core.py:2062-2226) -------------------------------------------------------------------
    def _compileModels(self):
        """Per transition model: its flat program and the op-aligned hyper-parameter values of all of its chains."""
        om = self.observationModel
        self._device = []
        for tm, hpv in zip(self.transitionModels, self.hyperParameterValues):
            self.setTransitionModel(tm, silent=True)
            if _tm_mod.needs_host_transition(tm):
                # (Study.fit / HyperStudy.fit apply such a model on the host between device steps; the batched online step has no
                #  host path: refuse loudly instead of running a built-in base class's program in place of the user's override)
                raise ConfigurationError('OnlineStudy: transition model {} defines computeForwardPrior / computeBackwardPrior itself (or '
                                         'contains such a model); user-defined transition models are supported by Study.fit and '
                                         'HyperStudy.fit only.'.format(type(tm).__name__))
            program = self._expandProgram(tm._program(om.parameterNames), 1)
            if len(self.gridSize) > 2 and not all(op[0] in (_abi.OP_GRW, _abi.OP_STATIC, _abi.OP_CHANGEPOINT) and op[4] < 0 for op in program):
                raise ConfigurationError('Observation models with more than two parameters can be combined with GaussianRandomWalk, '
                                         'Static and ChangePoint transition models.')
            # every step is a one-step problem resumed at t = -1 (core.py:2164-2165): the op values never change
            op_values = self._opValueMatrix(program, np.asarray(hpv, dtype=float) if len(hpv) > 0 else np.zeros((1, 0)),
                                            timestamps=[0.0], resume_time=-1.0)
            reset = self._changepointPrior() if any(op[0] == _abi.OP_CHANGEPOINT for op in program) else None
            indep = None
            if any(op[0] == _abi.OP_INDEPENDENT for op in program):
                indep = self._changepointPrior() / np.prod(self.latticeConstant)
            self._device.append(([(op[0], op[1], op[4], op[5]) for op in program], op_values, reset, indep))
            OnlineStudy._slot_counter[0] += 1
            self._slots.append(OnlineStudy._slot_counter[0])

    def _takeDataPoint(self, dataPoint):
        """The stream's bookkeeping of one call of ``step`` (reference core.py:2069-2096): a transition model that was only `set` joins
        the list of competing models; the first data point starts the raw series (after the duplicate-name and consistency checks), every
        later one is appended; time stamps count the data points from 0."""
        if self.tmCount is None:
            if self.transitionModel is None:
                raise ConfigurationError('No transition model set or added.')
            self.addTransitionModel('transition model', self.transitionModel)
        point = np.array(dataPoint if isinstance(dataPoint, list) else [dataPoint])
        if len(self.rawData) > 0:
            self.rawData = np.append(self.rawData, point, axis=0)
            self.rawTimestamps = np.append(self.rawTimestamps, self.rawTimestamps[-1] + 1)
            return
        print('+ Start model fit')
        names = list(flatten(self.hyperParameterNames))
        if len(set(names)) != len(names):
            raise ConfigurationError('Detected duplicate hyper-parameter names. Choose unique identifiers.')
        self.rawData = point
        Study._checkConsistency(self)
        self.rawTimestamps, self.formattedTimestamps = np.array([0]), []

    def step(self, dataPoint):
        """Update every chain with a new data point (float, int, or 1-D array for multi-dimensional data)."""
        self._takeDataPoint(dataPoint)

        om = self.observationModel
        if len(self.rawData) < om.segmentLength:
            print('+ Not enough data points to start analysis. Will wait for more data.')
            return
        self.formattedTimestamps.append(self.rawTimestamps[-1])
        eng = _engine_mod.get_engine()
        nTM = len(self.transitionModels)

        if self.firstStep:
            if not 1 <= len(self.gridSize) <= _abi.MAX_DIM:
                raise ConfigurationError('The MI355X engine supports observation models with 1 to {} parameters '
                                         '(got {}).'.format(_abi.MAX_DIM, len(self.gridSize)))
            self._prior = self._computePrior(silent=False)
            if self.transitionModelPrior is None:
                self.transitionModelPrior = np.ones(nTM) / nTM
                if nTM > 1:
                    print('    + Set flat transition model prior.')
            dV = np.prod(self.latticeConstant)
            self.logEvidenceList = [np.zeros(tmc) + np.log(dV) for tmc in self.tmCounts]
            self.hyperLogEvidenceList = np.array([0. for tmc in self.tmCounts])
            self.hyperParameterDistribution = [np.zeros(tmc) for tmc in self.tmCounts]
            self.transitionModelDistribution = np.zeros(nTM)
            self.localTransitionModelDistribution = np.zeros(nTM)
            self._compileModels()

        # the current data segment as a one-step problem; the likelihood is evaluated ONCE for all chains (core.py:2146):
        # in-kernel for the closed-form device models, on the host through the model's own pdf otherwise
        segment = self.rawData[-om.segmentLength:]
        data = np.asarray([segment], dtype=float)
        code = device_code(om)
        lik = None
        if code == _abi.OM_TABLE:
            lik = np.array([np.asarray(om.processedPdf(self.grid, segment), dtype=float) * np.ones(self.gridSize)])
        dV = np.prod(self.latticeConstant)

        for i, (ops, op_values, reset, indep) in enumerate(self._device):
            problem = FitProblem(obs_model=code, marginal=self.marginalGrid, lattice=self.latticeConstant, data=data,
                                 timestamps=np.asarray([self.rawTimestamps[-1]], dtype=float), prior=self._prior, ops=ops,
                                 reset_prior=reset, indep_prior=indep, lik=lik, seg_len=om.segmentLength,
                                 resume_time=-1.0,      # the reference evaluates transitions at len(formattedData) - 1 = -1
                                 carry_slot=self._slots[i])
            res = eng.fit(problem, op_values, evidence_only=True, resume=not self.firstStep, carry=True)
            with np.errstate(divide='ignore', invalid='ignore'):
                ni = res.local_evidence[:, 0] / dV                                               # core.py:2167
                self.logEvidenceList[i] = self.logEvidenceList[i] + np.log(ni)                   # :2170
                hpd = self.logEvidenceList[i] + np.log(self.hyperPriorValues[i])                 # :2171
                old = self.hyperLogEvidenceList[i]
                x = self.logEvidenceList[i] + np.log(self.hyperPriorValues[i])
                self.hyperLogEvidenceList[i] = _logsumexp(x)                                     # :2178 logsumexp
                self.transitionModelDistribution[i] = self.hyperLogEvidenceList[i]               # :2179
                self.localTransitionModelDistribution[i] = self.hyperLogEvidenceList[i] - old + \
                    np.log(self.transitionModelPrior[i])                                         # :2180-2181
                hpd = np.exp(hpd - np.amax(hpd))                                                 # :2184-2186
                hpd /= np.sum(hpd)
                if len(self.hyperGridConstants[i]) > 0:
                    hpd /= np.prod(self.hyperGridConstants[i])                                   # :2187-2188
            self.hyperParameterDistribution[i] = hpd

        with np.errstate(divide='ignore', invalid='ignore'):
            tmd = self.transitionModelDistribution
            tmd = np.exp(tmd - np.amax(tmd)); tmd /= np.sum(tmd)                                 # :2199-2201
            self.transitionModelDistribution = tmd
            ltd = self.localTransitionModelDistribution
            ltd = np.exp(ltd - np.amax(ltd)); ltd /= np.sum(ltd)                                 # :2203-2205
            self.localTransitionModelDistribution = ltd
            x = self.hyperLogEvidenceList + np.log(self.transitionModelPrior)
            m = np.amax(x)
            self.logEvidence = float(m + np.log(np.sum(np.exp(x - m))))                          # :2214

        # marginalised posterior = sum_i tmd_i sum_j hpd_ij prod(dh_i) posterior_ij, reduced on the device (:2194-2211)
        for i in range(nTM):
            w = self.hyperParameterDistribution[i] * np.prod(self.hyperGridConstants[i]) * self.transitionModelDistribution[i]
            eng.carry_mix(self._slots[i], w, accumulate=i > 0)
        self.marginalizedPosterior = eng.carry_read(0, -1, self.gridSize)

        if self.storeHistory:
            self.posteriorMeanValues.append(np.array([np.sum(self.marginalizedPosterior * g) for g in self.grid]))
            self.posteriorSequence.append(self.marginalizedPosterior.copy())
            self.hyperParameterSequence.append([h.copy() for h in self.hyperParameterDistribution])
            self.transitionModelSequence.append(self.transitionModelDistribution.copy())
            self.localTransitionModelSequence.append(self.localTransitionModelDistribution.copy())
        if self.firstStep:
            self.firstStep = False

    # ---- device-resident per-chain results, fetched on demand -----------------------------------------------------------
    @property
    def parameterPosterior(self):
        """[transition model][hyper-parameter value] filtered distributions (core.py:2173), copied from the device."""
        if self.firstStep:
            return None
        eng = _engine_mod.get_engine()
        return [np.array([eng.carry_read(s, j, self.gridSize) for j in range(c)]) for s, c in zip(self._slots, self.tmCounts)]

    @property
    def transitionModelPosterior(self):
        """Per transition model: posterior marginalised over its hyper-parameters (core.py:2190-2196)."""
        if self.firstStep:
            return None
        eng = _engine_mod.get_engine()
        out = []
        for i, s in enumerate(self._slots):
            eng.carry_mix(s, self.hyperParameterDistribution[i] * np.prod(self.hyperGridConstants[i]), accumulate=False)
            out.append(eng.carry_read(0, -1, self.gridSize))
        return np.array(out)

    # ---- accessors (reference core.py:2231-2835; plotting is not part of this build) ------------------------------------
    def _needHistory(self, what, alt):
        if not self.storeHistory:
            raise PostProcessingError('To get past {}, Online Study must be called with flag "storeHistory=True". '
                                      'Use "{}" instead.'.format(what, alt))

    @_plots('param_dist')
    def getParameterDistribution(self, t, name, plot=False, density=True, **kwargs):
        self._needHistory('parameter distributions', 'getCurrentParameterDistribution')
        k = self._parameterIndex(name)
        if t not in self.formattedTimestamps:
            raise PostProcessingError('Supplied time ({}) does not exist in data or is out of range.'.format(t))
        dist = self.posteriorSequence[list(self.formattedTimestamps).index(t)]
        axes = tuple(a for a in range(len(self.gridSize)) if a != k)
        marginal = np.sum(dist, axis=axes) if axes else np.array(dist)
        return self.marginalGrid[k], marginal / self.latticeConstant[k] if density else marginal

    @_plots('param_dist')
    def getCurrentParameterDistribution(self, name, plot=False, density=True, **kwargs):
        k = self._parameterIndex(name)
        axes = tuple(a for a in range(len(self.gridSize)) if a != k)
        marginal = np.sum(self.marginalizedPosterior, axis=axes) if axes else np.array(self.marginalizedPosterior)
        return self.marginalGrid[k], marginal / self.latticeConstant[k] if density else marginal

    def getCPD(self, name, plot=False, density=True, **kwargs):
        return self.getCurrentParameterDistribution(name, plot=plot, density=density, **kwargs)

    @_plots('param_dists')
    def getParameterDistributions(self, name, plot=False, density=True, **kwargs):
        self._needHistory('parameter distributions', 'getCurrentParameterDistribution')
        k = self._parameterIndex(name)
        post = np.array(self.posteriorSequence)
        axes = tuple(a + 1 for a in range(len(self.gridSize)) if a != k)
        marginal = np.sum(post, axis=axes) if axes else post
        return self.marginalGrid[k], marginal / self.latticeConstant[k] if density else marginal

    def getCurrentTransitionModelDistribution(self, local=False):
        d = self.localTransitionModelDistribution if local else self.transitionModelDistribution
        return np.array(self.transitionModelNames), d

    def getCTMD(self, local=False):
        return self.getCurrentTransitionModelDistribution(local=local)

    def getCurrentTransitionModelProbability(self, transitionModel, local=False):
        return self.getCurrentTransitionModelDistribution(local=local)[1][self.transitionModelNames.index(transitionModel)]

    def getCTMP(self, transitionModel, local=False):
        return self.getCurrentTransitionModelProbability(transitionModel, local=local)

    def getTransitionModelDistributions(self, local=False):
        self._needHistory('transition model distributions', 'getCurrentTransitionModelDistribution')
        seq = self.localTransitionModelSequence if local else self.transitionModelSequence
        return np.array(self.transitionModelNames), np.array(seq)

    def getTransitionModelProbabilities(self, transitionModel, local=False):
        return self.getTransitionModelDistributions(local=local)[1][:, self.transitionModelNames.index(transitionModel)]

    def getTMPs(self, transitionModel, local=False):
        return self.getTransitionModelProbabilities(transitionModel, local=local)

    def getCurrentParameterMeanValue(self, name):
        return np.sum(self.marginalizedPosterior * self.grid[self._parameterIndex(name)])

    def getParameterMeanValue(self, t, name):
        self._needHistory('parameter mean values', 'getCurrentParameterMeanValue')
        k = self._parameterIndex(name)
        if t not in self.formattedTimestamps:
            raise PostProcessingError('Supplied time ({}) does not exist in data or is out of range.'.format(t))
        return self.posteriorMeanValues[list(self.formattedTimestamps).index(t)][k]

    def getParameterMeanValues(self, name):
        self._needHistory('parameter mean values', 'getCurrentParameterMeanValue')
        return np.array(self.posteriorMeanValues).T[self._parameterIndex(name)]

    def _findHyperParameter(self, name):
        for i, tm in enumerate(self.transitionModels):
            if name in self.hyperParameterNames[i]:
                return i, list(self.hyperParameterNames[i]).index(name)
        raise PostProcessingError('No hyper-parameter "{}" found. Check hyper-parameter names.'.format(name))

    def getHyperParameterMeanValue(self, t, name):
        self._needHistory('hyper-parameter mean values', 'getCurrentHyperParameterMeanValue')
        i, k = self._findHyperParameter(name)
        if t not in self.formattedTimestamps:
            raise PostProcessingError('Supplied time ({}) does not exist in data or is out of range.'.format(t))
        h = self.hyperParameterSequence[list(self.formattedTimestamps).index(t)][i][:, None]
        return np.sum(self.hyperParameterValues[i] * h * np.prod(self.hyperGridConstants[i]), axis=0)[k]

    def getHyperParameterMeanValues(self, name):
        self._needHistory('hyper-parameter mean values', 'getCurrentHyperParameterMeanValue')
        i, k = self._findHyperParameter(name)
        seq = np.array([hp[i].tolist() for hp in self.hyperParameterSequence])[:, :, None]
        return np.sum(seq * self.hyperParameterValues[i] * np.prod(self.hyperGridConstants[i]), axis=1).T[k]

    def _marginalHyper(self, i, k, distribution):
        steps = [len(x) for x in self.allFlatHyperParameterValues[i]]
        d = np.asarray(distribution).reshape(steps, order='C')
        axes = tuple(a for a in range(len(steps)) if a != k)
        return np.sum(d, axis=axes) if axes else d

    @_plots('hyper_dist')
    def getHyperParameterDistribution(self, t, name, plot=False, **kwargs):
        self._needHistory('hyper-parameter distributions', 'getCurrentHyperParameterDistribution')
        i, k = self._findHyperParameter(name)
        if isinstance(t, str) and t == 'avg':
            h = np.sum([hp[i] for hp in self.hyperParameterSequence], axis=0) / len(self.hyperParameterSequence)
        else:
            if t not in self.formattedTimestamps:
                raise PostProcessingError('Supplied time ({}) does not exist in data or is out of range.'.format(t))
            h = self.hyperParameterSequence[list(self.formattedTimestamps).index(t)][i]
        return self.allFlatHyperParameterValues[i][k], self._marginalHyper(i, k, h)

    def getHPD(self, t, name, plot=False, **kwargs):
        return self.getHyperParameterDistribution(t, name, plot=plot, **kwargs)

    @_plots('hyper_dist')
    def getCurrentHyperParameterDistribution(self, name, plot=False, **kwargs):
        i, k = self._findHyperParameter(name)
        m = self._marginalHyper(i, k, self.hyperParameterDistribution[i]) * np.prod(self.hyperGridConstants[i])
        return self.allFlatHyperParameterValues[i][k], m

    def getCHPD(self, name, plot=False, **kwargs):
        return self.getCurrentHyperParameterDistribution(name, plot=plot, **kwargs)

    def getHyperParameterDistributions(self, name):
        self._needHistory('hyper-parameter distributions', 'getCurrentHyperParameterDistributions')
        i, k = self._findHyperParameter(name)
        seq = np.array([x[i] for x in self.hyperParameterSequence])
        values = np.array(self.hyperParameterValues[i])[:, k]
        unique = np.sort(np.unique(values))
        m = np.array([[np.sum(hp[values == v]) for hp in seq] for v in unique]).T
        return unique, m / np.sum(m, axis=1)[:, None]

    def getHPDs(self, name):
        return self.getHyperParameterDistributions(name)

    def plotParameterEvolution(self, name, color='b', gamma=0.5, **kwargs):
        """As :meth:`Study.plotParameterEvolution`, from the stored history (reference core.py:2359-2413)."""
        self._needHistory('parameter distributions', 'getCurrentParameterDistribution')
        return Study.plotParameterEvolution(self, name, color=color, gamma=gamma, **kwargs)

    def plotHyperParameterEvolution(self, name, color='b', gamma=0.5, **kwargs):
        """Image of a hyper-parameter's distribution over time with its mean values on top (reference core.py:2839-2898)."""
        from . import plotting
        values, dist = self.getHyperParameterDistributions(name)
        if len(values) > 1:
            step = values[1] - values[0]
            bounds = [values[0] - step / 2.0, values[-1] + step / 2.0]
        else:
            bounds = [values[0] - 0.5, values[0] + 0.5]
        plotting.evolution(self.formattedTimestamps, bounds, dist, self.getHyperParameterMeanValues(name), name,
                           color=color, gamma=gamma, **kwargs)
        plotting._plt().ylim(values[0], values[-1])

    def plot(self, name, **kwargs):
        """Evolution or distribution of a (hyper-)parameter, or the probability of a transition model over time
        (reference core.py:2900-2985)."""
        from . import plotting
        density = kwargs.pop('density', True)
        t = kwargs.pop('t', None)
        if t is not None and not self.storeHistory:
            raise PostProcessingError('Online study has only stored current parameter data ("storeHistory=False"), '
                                      'no time step can be specified, only current (hyper-)parameter distributions will'
                                      'be plotted.')
        hyper = any(name in names for names in self.hyperParameterNames)
        model = name in self.transitionModelNames
        if hyper and model:
            raise PostProcessingError('Duplicate names of hyper-/parameters/transition models, cannot use "plot" method.')
        if model:
            seq = self.localTransitionModelSequence if kwargs.pop('local', False) else self.transitionModelSequence
            plotting.line(self.formattedTimestamps, np.array(seq)[:, self.transitionModelNames.index(name)], **kwargs)
        elif hyper:
            if t is None and self.storeHistory:
                self.plotHyperParameterEvolution(name, color=kwargs.pop('color', 'b'), gamma=kwargs.pop('gamma', 0.5), **kwargs)
            elif t is not None:
                self.getHyperParameterDistribution(t, name, plot=True, **kwargs)
            else:
                self.getCurrentHyperParameterDistribution(name, plot=True, **kwargs)
        else:
            if t is None and self.storeHistory:
                self.plotParameterEvolution(name, color=kwargs.pop('color', 'b'), gamma=kwargs.pop('gamma', 0.5), **kwargs)
            elif t is not None:
                self.getParameterDistribution(t, name, plot=True, density=density, **kwargs)
            else:
                self.getCurrentParameterDistribution(name, plot=True, density=density, **kwargs)
