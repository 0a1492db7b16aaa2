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
    """Base class of all transition models."""

    hyperParameterNames = ()
    hyperParameterValues = ()

    def _program(self, parameterNames):
        """-> list of (op kind, axis, owner model, hyper-parameter index or None, serial segment or -1, flags)"""
        raise ConfigurationError('Transition model "{}" cannot be compiled for the MI355X engine.'.format(self))

    def computeForwardPrior(self, posterior, t):
        raise NotImplementedError('bayesloop_amd executes transition models inside the HIP step kernels; '
                                  'host-side computeForwardPrior is not part of this build.')

    def computeBackwardPrior(self, posterior, t):
        return self.computeForwardPrior(posterior, t - 1)


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

    def shifts(self, params, timestamps, resume_time=-1.0):
        """The 2 T values behind the DETERMINISTIC op (include/blhip.h): forward shift INTO step i, f(t'+1) - f(t') at the
        time stamp t' of step i-1 (entry 0: at ``resume_time``, used by OnlineStudy), then backward shift into step i,
        f(t'-1) - f(t') at the time stamp t' of step i+1 (reference transitionModels.py:573-577, :592-596)."""
        ts = np.asarray(timestamps, dtype=float)
        T = len(ts)
        f = lambda t: float(self.function(t - self.tOffset, **params))
        fwd = [f(resume_time + 1) - f(resume_time)] + [f(ts[i - 1] + 1) - f(ts[i - 1]) for i in range(1, T)]
        bwd = [f(ts[i + 1] - 1) - f(ts[i + 1]) for i in range(T - 1)] + [0.0]
        return np.array(fwd + bwd, dtype=float)


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

    def _program(self, parameterNames):
        program = []
        for seg, m in enumerate(self.models):
            for op in m._program(parameterNames):
                if op[4] != -1 or op[0] == _abi.OP_BREAKPOINT or (op[0] == _abi.OP_CHANGEPOINT and op[5] & 1):
                    raise ConfigurationError('Nested SerialTransitionModel instances are not supported.')
                if op[0] == _abi.OP_DETERMINISTIC:
                    raise ConfigurationError('A Deterministic model inside a SerialTransitionModel (time offset at the '
                                             'break-points, reference transitionModels.py:770-776) is not supported.')
                program.append((op[0], op[1], op[2], op[3], seg, op[5]))
        for k in range(len(self.hyperParameterNames)):
            if self.changePointMask[k]:
                program.append((_abi.OP_CHANGEPOINT, 0, self, k, -1, 1))
            else:
                program.append((_abi.OP_BREAKPOINT, 0, self, k, -1, 0))
        return program
