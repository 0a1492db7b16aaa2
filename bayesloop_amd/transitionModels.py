"""
Transition models -- the ``bl.tm`` namespace.

A transition model maps the posterior of one time step to the prior of the next (and the mirrored map backwards).
In the reference each model does this itself in numpy/SciPy (``computeForwardPrior`` / ``computeBackwardPrior``,
bayesloop/transitionModels.py:49-63, 96-118, 289-317, 632-662).  Here a model only DESCRIBES the map: ``fit()``
compiles the (possibly nested) model into a flat *transition program* -- a list of ops executed, in list order and in
both directions, inside the fused HIP step kernels:

    Static                  -> nothing                                   (identity)
    GaussianRandomWalk      -> GRW(axis, sigma): reflect-boundary Gaussian stencil along one grid axis
    ChangePoint             -> CHANGEPOINT(tChange): restart from the (re-normalised) prior at one time stamp
    RegimeSwitch            -> REGIMESWITCH(log10pMin): clamp from below and renormalise
    NotEqual                -> NOTEQUAL(log10pMin): max(p) - p, renormalise, clamp from below, renormalise
    Deterministic           -> DETERMINISTIC(axis) + 2 T DETERMINISTIC_ARG(shift per step, evaluated from the model's function)
    AlphaStableRandomWalk   -> ALPHASTABLE(axis, c) + ALPHASTABLE_ARG(alpha): zero-boundary stencil with the stable density, renormalised
    BivariateRandomWalk     -> BIVARIATE(sigma1) + 2 x BIVARIATE_ARG(sigma2, rho): dense 2-D convolution, zero boundary, renormalised
    Independent             -> INDEPENDENT: restart from the normalised prior at every step
    CombinedTransitionModel -> concatenation of the sub-models' programs
    SerialTransitionModel   -> the sub-models' programs tagged with their segment + BREAKPOINT / boundary CHANGEPOINT ops

Constructor arguments and the attributes the study classes rely on (``hyperParameterNames``, ``hyperParameterValues``,
``prior``, ``models``) are those of the reference, so the hyper-parameter plumbing of HyperStudy / ChangepointStudy
works unchanged.
"""
from __future__ import annotations

import numpy as np

from . import _abi
from .exceptions import ConfigurationError


class TransitionModel:
    """Base class of all transition models.

    The reference's plug-in boundary (bayesloop/transitionModels.py:49-63, called at core.py:411, :467, :2166) is the duck-typed pair
    ``computeForwardPrior(posterior, t)`` / ``computeBackwardPrior(posterior, t)``.  Both sides of it exist here:

    * the built-in models below answer these calls by running their OWN one-step transition program on the GPU on the distribution
      they are handed (:func:`_device_transition`), so code written against the reference -- a user-defined model that wraps or
      combines built-in ones, a notebook that calls ``tm.computeForwardPrior(p, t)`` -- keeps working;
    * a model the library has no program for (a subclass that defines ``computeForwardPrior`` itself, exactly as the reference
      documents) is fitted by ``Study.fit`` with the transition applied by THAT method on the host between device steps
      (``Study._fitHostTransition``: likelihood products, normalisations, sums and means stay on the GPU).
    """

    hyperParameterNames = ()
    hyperParameterValues = ()

    def _program(self, parameterNames):
        """-> list of (op kind, axis, owner model, hyper-parameter index or None, serial segment or -1, flags); None: the library has
        no program for this model (user-defined: Study.fit calls its computeForwardPrior / computeBackwardPrior on the host)"""
        return None

    def computeForwardPrior(self, posterior, t):
        return _device_transition(self, posterior, t)

    def computeBackwardPrior(self, posterior, t):
        return self.computeForwardPrior(posterior, t - 1)


_OWN_MODULE = __name__
_APPLY_SLOT = 0x7fff0001          # the carried-state slot of the library context the one-step applications go through


def needs_host_transition(model):
    """True if ``model`` (or a sub-model) has to be applied by its own computeForwardPrior / computeBackwardPrior on the host: it is
    not one of the built-in classes, or it is a subclass of one that overrides either method."""
    for name in ('computeForwardPrior', 'computeBackwardPrior'):
        fn = getattr(type(model), name, None)
        if fn is None or getattr(fn, '__module__', None) != _OWN_MODULE:
            return True
    prog = getattr(type(model), '_program', None)
    if prog is None or getattr(prog, '__module__', None) != _OWN_MODULE or prog is TransitionModel._program:
        return True
    return any(needs_host_transition(m) for m in getattr(model, 'models', ()))


def _scalar(model, k):
    v = model.hyperParameterValues[k]
    if isinstance(v, str) or np.ndim(v) != 0:
        raise ConfigurationError('Hyper-parameter "{}" holds several values; computeForwardPrior needs one value per '
                                 'hyper-parameter.'.format(model.hyperParameterNames[k]))
    return float(v)


def _device_transition(model, posterior, t, shift=None):
    """``model``'s transition of the distribution ``posterior`` at time stamp ``t`` (reference semantics, any input scale), computed
    on the GPU: the normalised input goes into a carried-state slot of the library context, ONE resumed forward step with a flat
    likelihood applies the model's own op program to it (the machinery of OnlineStudy.step: BLHIP_RESUME | BLHIP_CARRY), the
    result comes back with its sum.  Leaf models only (CombinedTransitionModel / SerialTransitionModel delegate like the
    reference does).  ``shift``: Deterministic's grid shift for this call (its backward shift is not a forward shift at t - 1)."""
    from . import engine as _engine_mod
    from .engine import FitProblem
    study = getattr(model, 'study', None)
    if study is None or not getattr(study, 'gridSize', None):
        raise ConfigurationError('Transition model "{}" is not attached to a study with an observation model '
                                 '(Study.setTransitionModel sets model.study).'.format(model))
    x = np.asarray(posterior, dtype=float)
    grid_size = list(study.gridSize)
    if list(x.shape) != grid_size:
        raise ConfigurationError('computeForwardPrior: distribution of shape {} on a grid of shape {}.'.format(list(x.shape), grid_size))
    if len(grid_size) > 2:
        raise NotImplementedError('computeForwardPrior on the device is limited to grids with one or two parameters.')
    program = model._program(study.observationModel.parameterNames)
    if program is None:
        raise NotImplementedError('Transition model "{}" has no device program; define computeForwardPrior.'.format(model))
    s = float(np.sum(x))
    if not (s > 0.0 and np.isfinite(s)):
        raise ConfigurationError('computeForwardPrior: the distribution has no positive finite mass.')
    program = study._expandProgram(program, 1)
    params = {n: _scalar(model, k) for k, n in enumerate(model.hyperParameterNames)}
    values = np.full((1, max(1, len(program))), np.nan)
    linear = True                    # homogeneous of degree 1 (the result scales with the input) or normalised output?
    for j, (kind, axis, owner, k, seg, flg) in enumerate(program):
        if kind not in (_abi.OP_GRW, _abi.OP_STATIC):
            linear = False
        if isinstance(k, tuple):                             # ('shift', q) of a Deterministic model: [forward into step 0, backward]
            values[0, j] = (owner.shifts(params, [t + 1.0], float(t))[k[1]] if shift is None else shift) if k[1] == 0 else 0.0
        elif k is not None:
            values[0, j] = _scalar(owner, k)
            if kind == _abi.OP_REGIMESWITCH:                  # clamp at 10**v dV of the RAW input = 10**(v - log10 s) dV of x / s
                values[0, j] -= np.log10(s)
    eng = _engine_mod.get_engine()
    dV = float(np.prod(study.latticeConstant))
    reset = study._changepointPrior() if any(op[0] == _abi.OP_CHANGEPOINT for op in program) else None
    indep = study._changepointPrior() / dV if any(op[0] == _abi.OP_INDEPENDENT for op in program) else None
    eng.carry_write(_APPLY_SLOT, (x / s).reshape([1] + grid_size))
    problem = FitProblem(obs_model=_abi.OM_TABLE, marginal=study.marginalGrid, lattice=study.latticeConstant,
                         data=np.zeros((1, 1)), timestamps=np.asarray([t + 1.0]), prior=x / s,
                         ops=[(op[0], op[1], op[4], op[5]) for op in program], reset_prior=reset, indep_prior=indep,
                         lik=np.ones([1] + grid_size), seg_len=1, resume_time=float(t), carry_slot=_APPLY_SLOT)
    res = eng.fit(problem, values, evidence_only=True, resume=True, carry=True)
    norm = float(res.local_evidence[0, 0]) / dV              # sum of T(x / s) before the step's normalisation
    out = eng.carry_read(_APPLY_SLOT, 0, grid_size)
    return out * (norm * s if linear else norm)


def _as_values(value):
    return np.array(value) if isinstance(value, (list, tuple)) else value


class Static(TransitionModel):
    """Constant parameters (reference transitionModels.py:34-63)."""

    def __init__(self):
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = []
        self.hyperParameterValues = []
        self.prior = None
        self.tOffset = 0

    def __str__(self):
        return 'Static/constant parameter values'

    def computeForwardPrior(self, posterior, t):
        return posterior                      # (the same array object, as in the reference: transitionModels.py:49-60)

    def _program(self, parameterNames):
        return [(_abi.OP_STATIC, 0, self, None, -1, 0)]


class GaussianRandomWalk(TransitionModel):
    """Gaussian fluctuations of one parameter with standard deviation sigma (reference transitionModels.py:66-118)."""

    def __init__(self, name='sigma', value=None, target=None, prior=None):
        if target is None:
            raise ConfigurationError('No parameter set for transition model "GaussianRandomWalk"')
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = [name]
        self.hyperParameterValues = [_as_values(value)]
        self.prior = prior
        self.selectedParameter = target
        self.tOffset = 0

    def __str__(self):
        return 'Gaussian random walk'

    def _program(self, parameterNames):
        if self.selectedParameter not in parameterNames:
            raise ConfigurationError('GaussianRandomWalk: observation model has no parameter "{}".'
                                     .format(self.selectedParameter))
        return [(_abi.OP_GRW, list(parameterNames).index(self.selectedParameter), self, 0, -1, 0)]


class ChangePoint(TransitionModel):
    """Abrupt change right after time stamp tChange (reference transitionModels.py:263-317)."""

    def __init__(self, name='tChange', value=None, prior=None):
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = [name]
        self.hyperParameterValues = [_as_values(value)]
        self.prior = prior
        self.tOffset = 0

    def __str__(self):
        return 'Change-point'

    def computeForwardPrior(self, posterior, t):
        """reference transitionModels.py:289-314: at t == tChange the (re-normalised) prior times the cell volume -- the array the
        device kernels restart from (Study._changepointPrior) -- else the distribution itself.  No arithmetic on ``posterior``."""
        if t == _scalar(self, 0):
            return self.study._changepointPrior()
        return posterior

    def _program(self, parameterNames):
        return [(_abi.OP_CHANGEPOINT, 0, self, 0, -1, 0)]


class CombinedTransitionModel(TransitionModel):
    """Several models acting at the same time, applied in the given order (reference transitionModels.py:609-662)."""

    def __init__(self, *args):
        if any(str(arg) == 'Break-point' for arg in args):
            raise ConfigurationError('The "BreakPoint" transition model can only be used with the '
                                     '"SerialTransitionModel" class.')
        self.study = None
        self.latticeConstant = None
        self.models = args
        self.tOffset = 0

    def __str__(self):
        return 'Combined transition model'

    def _propagate(self, m):
        m.latticeConstant = self.latticeConstant      # reference transitionModels.py:646-648
        m.study = self.study
        m.tOffset = self.tOffset

    def computeForwardPrior(self, posterior, t):
        """reference transitionModels.py:632-649: the sub-models one after the other, in list order (each one through its own
        computeForwardPrior: built-in ones on the device, user-defined ones on the host)"""
        newPrior = np.array(posterior, dtype=float)
        for m in self.models:
            self._propagate(m)
            newPrior = m.computeForwardPrior(newPrior, t)
        return newPrior

    def computeBackwardPrior(self, posterior, t):
        """reference transitionModels.py:651-662: the SAME list order backwards"""
        newPrior = np.array(posterior, dtype=float)
        for m in self.models:
            self._propagate(m)
            newPrior = m.computeBackwardPrior(newPrior, t)
        return newPrior

    def _program(self, parameterNames):
        program = []
        for m in self.models:
            program += m._program(parameterNames)
        return program


class RegimeSwitch(TransitionModel):
    """Minimal probability density 10**log10pMin for every parameter value at each step (clamp from below, then
    renormalise; reference transitionModels.py:366-415)."""

    def __init__(self, name='log10pMin', value=None, prior=None):
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = [name]
        self.hyperParameterValues = [_as_values(value)]
        self.prior = prior
        self.tOffset = 0

    def __str__(self):
        return 'Regime-switching model'

    def _program(self, parameterNames):
        return [(_abi.OP_REGIMESWITCH, 0, self, 0, -1, 0)]


class NotEqual(TransitionModel):
    """Unlikely parameter values are preferred in the next step: max(p) - p, renormalised, clamped from below at
    10**log10pMin and renormalised again (reference transitionModels.py:418-474; mostly used with OnlineStudy)."""

    def __init__(self, name='log10pMin', value=None, prior=None):
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = [name]
        self.hyperParameterValues = [_as_values(value)]
        self.prior = prior
        self.tOffset = 0

    def __str__(self):
        return 'Not-Equal model'

    def _program(self, parameterNames):
        return [(_abi.OP_NOTEQUAL, 0, self, 0, -1, 0)]


class Deterministic(TransitionModel):
    """Deterministic parameter variation: the distribution of ``target`` is shifted from step to step by the increments
    of ``function(t, **hyperParameters)`` (cubic-spline shift, edge-extended, renormalised; reference
    transitionModels.py:477-606).  The keyword arguments of ``function`` are the hyper-parameters, their defaults the
    hyper-parameter values."""

    def __init__(self, function=None, target=None, prior=None):
        import inspect
        self.study = None
        self.latticeConstant = None
        self.function = function
        self.selectedParameter = target
        self.tOffset = 0
        if target is None:
            raise ConfigurationError('No parameter set for transition model "Deterministic"')
        spec = inspect.getfullargspec(self.function)
        defaults = spec.defaults or ()
        if not len(spec.args) == len(defaults) + 1:
            raise ConfigurationError('Function to define deterministic transition model can only contain one '
                                     'non-keyword argument (time; first argument) and keyword-arguments '
                                     '(hyper-parameters) with default values.')
        self.hyperParameterNames = list(spec.args[1:])
        self.hyperParameterValues = [_as_values(d) for d in defaults]
        if prior is None:
            self.prior = [None] * len(defaults)
        elif isinstance(prior, (list, tuple)):
            if len(prior) != len(defaults):
                raise ConfigurationError('{} priors are defined for transition model "{}", but model contains {} '
                                         'hyper-parameters.'.format(len(prior), self.function.__name__, len(defaults)))
            self.prior = list(prior)
        else:
            self.prior = [prior]

    def __str__(self):
        return 'Deterministic model ({})'.format(self.function.__name__)

    def _program(self, parameterNames):
        if self.selectedParameter not in parameterNames:
            raise ConfigurationError('Deterministic: observation model has no parameter "{}".'.format(self.selectedParameter))
        # the 2 T per-step shifts follow this op as DETERMINISTIC_ARG ops (added when the study is compiled, core.py)
        return [(_abi.OP_DETERMINISTIC, list(parameterNames).index(self.selectedParameter), self, None, -1, 0)]

    def computeBackwardPrior(self, posterior, t):
        """reference transitionModels.py:586-606: shifted by f(t - 1) - f(t) (NOT the forward shift at t - 1)"""
        params = {n: _scalar(self, k) for k, n in enumerate(self.hyperParameterNames)}
        d = float(self.function(t - 1 - self.tOffset, **params)) - float(self.function(t - self.tOffset, **params))
        return _device_transition(self, posterior, t - 1, shift=d)

    def shifts(self, params, timestamps, resume_time=-1.0, t_offset=None):
        """The 2 T values behind the DETERMINISTIC op (include/blhip.h): forward shift INTO step i, f(t'+1) - f(t') at the
        time stamp t' of step i-1 (entry 0: at ``resume_time``, used by OnlineStudy), then backward shift into step i,
        f(t'-1) - f(t') at the time stamp t' of step i+1 (reference transitionModels.py:573-577, :592-596).  ``t_offset``: the time
        the model counts from (inside a SerialTransitionModel: the break-point that starts its segment, :770-776)."""
        ts = np.asarray(timestamps, dtype=float)
        T = len(ts)
        off = self.tOffset if t_offset is None else t_offset
        # the 2 T + 2 time stamps the function is needed at, in one call where the user's function takes arrays (a hyper-study over
        # break-points evaluates this per chain: tens of thousands of chains)
        at = np.concatenate(([resume_time + 1, resume_time], ts[:-1] + 1, ts[:-1], ts[1:] - 1, ts[1:])) - off
        try:
            v = np.asarray(self.function(at, **params), dtype=float)
            if v.shape != at.shape:
                raise ValueError
            # a function that accepts arrays need not be elementwise (np.cumsum, t.mean(), t[0] ...): the reference calls it per
            # scalar (transitionModels.py:573-577), so the array result must agree with scalar calls -- checked at three entries
            for i in {0, len(at) // 2, len(at) - 1}:
                if not np.array_equal(v[i], float(self.function(at[i], **params)), equal_nan=True):
                    raise ValueError
        except Exception:                        # noqa: BLE001 -- a function written for scalars only, or not elementwise
            v = np.array([float(self.function(a, **params)) for a in at])
        n = T - 1
        fwd = np.concatenate(([v[0] - v[1]], v[2:2 + n] - v[2 + n:2 + 2 * n]))
        bwd = np.concatenate((v[2 + 2 * n:2 + 3 * n] - v[2 + 3 * n:2 + 4 * n], [0.0]))
        return np.concatenate((fwd, bwd))

    def shifts_many(self, names, rows, timestamps, resume_time=-1.0, t_offsets=None):
        """``shifts`` for many parameter sets at once: ``rows`` (U, len(names)) hyper-parameter values, ``t_offsets`` (U,) or None ->
        (U, 2 T).  One broadcast call of the user's function where it takes arrays in every argument (checked against the per-set
        evaluation on the first and the last row: a function that does not broadcast falls back to one call per set)."""
        rows = np.asarray(rows, dtype=float).reshape(len(rows), len(names))
        U = rows.shape[0]
        one = lambda u: self.shifts(dict(zip(names, rows[u])), timestamps, resume_time, t_offset=None if t_offsets is None else t_offsets[u])
        if U <= 2:
            return np.array([one(u) for u in range(U)])
        ts = np.asarray(timestamps, dtype=float)
        T = len(ts)
        n = T - 1
        off = np.full((U, 1), float(self.tOffset)) if t_offsets is None else np.asarray(t_offsets, dtype=float).reshape(U, 1)
        at = np.concatenate(([resume_time + 1, resume_time], ts[:-1] + 1, ts[:-1], ts[1:] - 1, ts[1:]))[None, :] - off
        try:
            with np.errstate(all='ignore'):
                v = np.asarray(self.function(at, **{nm: rows[:, [i]] for i, nm in enumerate(names)}), dtype=float)
            if v.shape != at.shape:
                raise ValueError
            out = np.concatenate((v[:, [0]] - v[:, [1]], v[:, 2:2 + n] - v[:, 2 + n:2 + 2 * n],
                                  v[:, 2 + 2 * n:2 + 3 * n] - v[:, 2 + 3 * n:2 + 4 * n], np.zeros((U, 1))), axis=1)
            for u in (0, U - 1):
                if not np.array_equal(out[u], one(u), equal_nan=True):
                    raise ValueError
            return out
        except Exception:                        # noqa: BLE001 -- a function that does not broadcast over its parameters
            return np.array([one(u) for u in range(U)])


class AlphaStableRandomWalk(TransitionModel):
    """Heavy-tailed fluctuations of one parameter: convolution with a symmetric alpha-stable density of scale c and tail
    index alpha (alpha = 1: Cauchy, 2: Gauss), zero boundary, renormalised (reference transitionModels.py:121-260)."""

    def __init__(self, name1='c', value1=None, name2='alpha', value2=None, target=None, prior=(None, None)):
        if target is None:
            raise ConfigurationError('No parameter set for transition model "AlphaStableRandomWalk"')
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = [name1, name2]
        self.hyperParameterValues = [_as_values(value1), _as_values(value2)]
        self.prior = prior
        self.selectedParameter = target
        self.tOffset = 0

    def __str__(self):
        return 'Alpha-stable random walk'

    def _program(self, parameterNames):
        if self.selectedParameter not in parameterNames:
            raise ConfigurationError('AlphaStableRandomWalk: observation model has no parameter "{}".'
                                     .format(self.selectedParameter))
        axis = list(parameterNames).index(self.selectedParameter)
        return [(_abi.OP_ALPHASTABLE, axis, self, 0, -1, 0), (_abi.OP_ALPHASTABLE_ARG, 0, self, 1, -1, 0)]


class BivariateRandomWalk(TransitionModel):
    """Correlated Gaussian fluctuations of both parameters of a two-parameter observation model: dense 2-D convolution
    with a bivariate normal kernel, zero boundary, renormalised (reference transitionModels.py:843-911)."""

    def __init__(self, name1='sigma1', value1=None, name2='sigma2', value2=None, name3='rho', value3=None,
                 prior=(None, None, None)):
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = [name1, name2, name3]
        self.hyperParameterValues = [_as_values(value1), _as_values(value2), _as_values(value3)]
        self.prior = prior
        self.tOffset = 0

    def __str__(self):
        return 'Bivariate random walk'

    def _program(self, parameterNames):
        if len(parameterNames) != 2:
            raise ConfigurationError('BivariateRandomWalk needs an observation model with exactly two parameters.')
        return [(_abi.OP_BIVARIATE, 0, self, 0, -1, 0), (_abi.OP_BIVARIATE_ARG, 0, self, 1, -1, 0),
                (_abi.OP_BIVARIATE_ARG, 0, self, 2, -1, 0)]


class Independent(TransitionModel):
    """Independent observations: the (normalised) prior is restored at every step (reference transitionModels.py:320-363)."""

    def __init__(self):
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames = []
        self.hyperParameterValues = []
        self.prior = None
        self.tOffset = 0

    def __str__(self):
        return 'Independent observations model'

    def computeForwardPrior(self, posterior, t):
        """reference transitionModels.py:339-360: the normalised prior, whatever the posterior"""
        return self.study._changepointPrior() / np.prod(self.study.latticeConstant)

    def _program(self, parameterNames):
        return [(_abi.OP_INDEPENDENT, 0, self, None, -1, 0)]


class BreakPoint(TransitionModel):
    """Break-point between two sub-models of a SerialTransitionModel (reference transitionModels.py:821-840)."""

    def __init__(self, name='tBreak', value=None, prior=None):
        self.name = name
        self.value = _as_values(value)
        self.prior = prior

    def __str__(self):
        return 'Break-point'


class SerialTransitionModel(TransitionModel):
    """Different models act at different times: n sub-models separated by n-1 break-points / change-points given in
    increasing order (reference transitionModels.py:665-818).  At time stamp t the sub-model with index
    ``#(boundaries <= t)`` acts; a change-point additionally restarts the parameters from the prior."""

    def __init__(self, *args):
        self.study = None
        self.latticeConstant = None
        self.hyperParameterNames, self.hyperParameterValues, self.prior = [], [], []
        self.models, mask = [], []
        for arg in args:
            if str(arg) == 'Break-point':
                self.hyperParameterNames.append(arg.name)
                self.hyperParameterValues.append(arg.value)
                self.prior.append(arg.prior)
                mask.append(0)
            elif str(arg) == 'Change-point':
                self.hyperParameterNames.append(arg.hyperParameterNames[0])
                self.hyperParameterValues.append(arg.hyperParameterValues[0])
                self.prior.append(arg.prior)
                mask.append(1)
            else:
                self.models.append(arg)
        self.changePointMask = np.array(mask).astype(bool)

        first = []
        for v in self.hyperParameterValues:
            first.append(v if (isinstance(v, str) or np.ndim(v) == 0) else v[0])
        for x, y in zip(first, first[1:]):
            if not (isinstance(x, str) or isinstance(y, str)) and not x < y:
                raise ConfigurationError('Time steps for structural breaks and/or change-points have to be passed in '
                                         'monotonically increasing order.')
        if len(self.models) - 1 != len(self.hyperParameterValues):
            raise ConfigurationError('Wrong number of structural breaks/change-points and models. For n models, n-1 '
                                     'structural breaks/change-points are required.')

    def __str__(self):
        return 'Serial transition model'

    def _active(self, t):
        """the sub-model acting at time stamp t = number of break times <= t (reference transitionModels.py:767-773)"""
        values = [_scalar(self, k) for k in range(len(self.hyperParameterValues))]
        k = int(np.sum(np.array(values) <= t)) if values else 0
        m = self.models[k]
        m.latticeConstant = self.latticeConstant
        m.study = self.study
        m.tOffset = values[k - 1] if k > 0 else 0
        return m, values

    def _changePointCheck(self, distribution, t, values):
        """reference transitionModels.py:789-815: a change-point boundary at t restarts from the prior"""
        if len(values) and t in np.array(values)[self.changePointMask]:
            return self.study._changepointPrior()
        return distribution

    def computeForwardPrior(self, posterior, t):
        m, values = self._active(t)
        return self._changePointCheck(m.computeForwardPrior(posterior, t), t, values)

    def computeBackwardPrior(self, posterior, t):
        m, values = self._active(t - 1)
        return self._changePointCheck(m.computeBackwardPrior(posterior, t), t - 1, values)

    def _program(self, parameterNames):
        program = []
        for seg, m in enumerate(self.models):
            for op in m._program(parameterNames):
                if op[4] != -1 or op[0] == _abi.OP_BREAKPOINT or (op[0] == _abi.OP_CHANGEPOINT and op[5] & 1):
                    raise ConfigurationError('Nested SerialTransitionModel instances are not supported.')
                # (a Deterministic sub-model counts its time from the break-point that starts its segment -- reference
                #  transitionModels.py:770-776 sets tOffset -- which Study._opValueMatrix applies per chain when it evaluates the shifts)
                program.append((op[0], op[1], op[2], op[3], seg, op[5]))
        for k in range(len(self.hyperParameterNames)):
            if self.changePointMask[k]:
                program.append((_abi.OP_CHANGEPOINT, 0, self, k, -1, 1))
            else:
                program.append((_abi.OP_BREAKPOINT, 0, self, k, -1, 0))
        return program
