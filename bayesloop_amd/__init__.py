"""
bayesloop_amd -- the grid-based forward-backward inference loop of bayesloop (``Study.fit`` / ``HyperStudy.fit`` /
``ChangepointStudy.fit``) on AMD MI355X (gfx950): Python host code over hand-written fp64 HIP kernels, called through
the C-ABI of ``libblhip.so`` (``include/blhip.h``).  Same public surface as the reference for this path:

    import bayesloop_amd as bl
    S = bl.Study(); S.loadExampleData()
    S.set(bl.om.Poisson('rate', bl.oint(0, 6, 1000)), bl.tm.GaussianRandomWalk('sigma', 0.2, target='rate'))
    S.fit()
"""
from .core import Study, HyperStudy, ChangepointStudy, OnlineStudy
from . import observationModels
from . import observationModels as om
from . import transitionModels
from . import transitionModels as tm
from .helper import cint, oint
from .fileIO import save, load
from .exceptions import ConfigurationError, PostProcessingError, BackendError
from . import dist
from .engine import get_engine, set_engine

__version__ = '0.1.0'

# Parser (scipy.special: 0.2 - 0.4 s of import on its own) and the Jeffreys-prior helpers are not on the fit() path: first use imports them
_LAZY = {'Parser': ('.parser', 'Parser'), 'parser': ('.parser', None), 'jeffreys': ('.jeffreys', None),
         'getJeffreysPrior': ('.jeffreys', 'getJeffreysPrior'), 'computeJeffreysPriorAR1': ('.jeffreys', 'computeJeffreysPriorAR1')}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        m = importlib.import_module(mod, __name__)
        v = m if attr is None else getattr(m, attr)
        globals()[name] = v
        return v
    raise AttributeError('module %r has no attribute %r' % (__name__, name))


def __dir__():
    return sorted(list(globals()) + list(_LAZY))
