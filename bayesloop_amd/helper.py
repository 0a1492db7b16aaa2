"""Grid constructors and nested-list utilities (API parity with bayesloop/helper.py:11-62, 90-120 of the reference)."""
import numpy as np


def oint(start, stop, num):
    """``num`` evenly spaced values of the OPEN interval (start, stop) (reference helper.py:90-104)."""
    return np.linspace(start, stop, num + 2)[1:-1]


def cint(start, stop, num):
    """``num`` evenly spaced values of the CLOSED interval [start, stop] (reference helper.py:107-120)."""
    return np.linspace(start, stop, num)


def flatten(lst):
    """Depth-first generator over an arbitrarily nested list/tuple (reference helper.py:47-62)."""
    for item in lst:
        if isinstance(item, (list, tuple)):
            yield from flatten(item)
        else:
            yield item


def recursiveIndex(nestedList, query):
    """Index path of the first occurrence of ``query`` in a nested list, [] if absent (reference helper.py:26-44)."""
    for k, element in enumerate(nestedList):
        if isinstance(element, (list, tuple)):
            path = recursiveIndex(element, query)
            if path:
                return [k] + path
        if not isinstance(element, (list, tuple, np.ndarray)) and element == query:
            return [k]
    return []


def assignNestedItem(lst, index, value):
    """In-place assignment into a nested list by index path (reference helper.py:11-23)."""
    target = lst
    for k in index[:-1]:
        target = target[k]
    target[index[-1]] = value


def createColormap(color, min_factor=1.0, max_factor=0.95):
    """Colormap from white (gray level ``min_factor``) to ``max_factor`` x ``color`` (reference helper.py:65-87)."""
    from .plotting import light_colormap
    return light_colormap(color, min_factor=min_factor, max_factor=max_factor)
