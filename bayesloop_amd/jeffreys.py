"""
Jeffreys priors outside the built-in observation models (reference bayesloop/jeffreys.py; exported as ``bl.getJeffreysPrior``
and ``bl.computeJeffreysPriorAR1``).
"""
import numpy as np

from .exceptions import ConfigurationError
from .observationModels import jeffreys_prior_of


def getJeffreysPrior(rv):
    """Symbolic Jeffreys prior sqrt(det I) of a SymPy random variable: ``(expression, numpy lambda of its parameters)``
    (reference jeffreys.py:17-68)."""
    return jeffreys_prior_of(rv)


def computeJeffreysPriorAR1(study, t=1):
    """Jeffreys prior of the stationary AR(1) process with the exact likelihood (H. Uhlig, Econometric Theory 10 (1994),
    eq. 31), on the grid of a study whose observation model is ``AR1`` or ``ScaledAR1`` (reference jeffreys.py:71-108):

        p(rho, s) ~ s^-2 exp(-d0^2 (1 - rho^2) / (2 s^2)) sqrt(4 rho^2 / (1 - rho^2) + 2 (n + 1))

    with ``d0`` = the observation before time step ``t``, ``n`` = number of data points, and ``s = sigma sqrt(1 - rho^2)`` for the
    scaled model.  Returns the array normalised to sum 1."""
    kind = type(study.observationModel).__name__
    if kind not in ('AR1', 'ScaledAR1'):
        raise ConfigurationError('Jeffreys prior for autoregressive process can only be used with AR1 and ScaledAR1 models.')
    rho, s = (np.asarray(a, dtype=float) for a in study.grid)
    if kind == 'ScaledAR1':
        s = s * np.sqrt(1.0 - rho ** 2)
    if np.any(np.abs(rho) >= 1.0):
        raise ConfigurationError('Jeffreys prior for auto-regressive process is only implemented for stationary processes. '
                                 'Values abs(r) >= 1 are not allowed for this implementation of the prior.')
    if len(study.rawData) == 0:
        raise ConfigurationError('Data must be loaded before computing the Jeffreys prior for the autoregressive process.')
    d0 = float(np.asarray(study.rawData)[t - 1])
    n = len(study.rawData)
    prior = np.exp(-d0 ** 2 * (1.0 - rho ** 2) / (2.0 * s ** 2)) / s ** 2 * np.sqrt(4.0 * rho ** 2 / (1.0 - rho ** 2) + 2.0 * (n + 1))
    return prior / np.sum(prior)
