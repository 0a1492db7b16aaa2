"""``bl.save`` / ``bl.load``: study objects to and from disk (reference bayesloop/fileIO.py:10-37, dill-based so that
lambda priors and user-defined observation models survive).  Results that still live on the GPU are copied to the host
first (``Study.__getstate__``); device handles are never pickled."""
from __future__ import annotations


def _pickler():
    try:
        import dill
        return dill
    except ImportError:            # plain pickle works for studies without lambdas / local functions
        import pickle
        return pickle


def save(filename, study):
    """Save an instance of a study class to file."""
    p = _pickler()
    with open(filename, 'wb') as f:
        p.dump(study, f, protocol=p.HIGHEST_PROTOCOL)
    print('+ Successfully saved current study.')


def load(filename):
    """Load an instance of a study class saved with :func:`save`."""
    p = _pickler()
    with open(filename, 'rb') as f:
        study = p.load(f)
    print('+ Successfully loaded study.')
    return study
