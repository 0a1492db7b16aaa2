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
from .parser import Parser
from . import jeffreys
from .jeffreys import getJeffreysPrior, computeJeffreysPriorAR1
from .exceptions import ConfigurationError, PostProcessingError, BackendError
from . import dist
from .engine import get_engine, set_engine

__version__ = '0.1.0'
