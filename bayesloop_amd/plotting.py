"""
Matplotlib views of the results (reference: the ``plot=True`` branches of the accessors and ``plot`` / ``plotParameterEvolution``
/ ``plotHyperParameterEvolution``, bayesloop/core.py:920-927, 985-998, 1007-1096, 1572-1584, 1660-1687, 1702-1741, 1916-1923,
2307-2315, 2359-2413, 2703-2718, 2762-2777, 2839-2985).  Same figures, own code; matplotlib is imported on first use so that the
compute path does not depend on it.  Every evolution image is built from the (T, n) marginals -- reduced on the GPU while the
posterior sequence still lives there -- never from the (T, G) sequence.
"""
import numpy as np


def _plt():
    import matplotlib.pyplot as plt
    return plt


def light_colormap(color, min_factor=1.0, max_factor=0.95):
    """Gray level ``min_factor`` (white by default) -> ``max_factor`` x ``color`` (the reference's helper.createColormap,
    helper.py:65-87)."""
    from matplotlib.colors import LinearSegmentedColormap, to_rgb
    top = tuple(max_factor * c for c in to_rgb(color))
    return LinearSegmentedColormap.from_list('bl_' + str(color), [(min_factor,) * 3, top])


def _is_regular(x):
    x = np.asarray(x, dtype=float)
    return len(x) < 3 or not np.any(np.abs(np.diff(np.diff(x))) > 1e-10)


def distribution(x, p, xlabel, density=True, **kwargs):
    plt = _plt()
    plt.fill_between(x, 0, p, **kwargs)
    plt.xlabel(xlabel)
    plt.ylabel('probability density' if density else 'probability')


def marginal_image(stamps, bounds, marginals, color='b', gamma=1.0):
    plt = _plt()
    plt.imshow(np.asarray(marginals).T ** gamma, origin='lower', cmap=light_colormap(color),
               extent=[stamps[0], stamps[-1]] + list(bounds), aspect='auto')


def evolution(stamps, bounds, marginals, means, ylabel, color='b', gamma=0.5, **kwargs):
    """Gamma-corrected image of the marginals over time with the mean values on top."""
    plt = _plt()
    stamps = np.asarray(stamps, dtype=float)
    if len(stamps) > 2 and not np.all(np.diff(stamps) == np.diff(stamps)[0]):
        print('! WARNING: Time stamps are not equally spaced. This may result in false plotting of parameter distributions.')
    m = np.array(marginals, dtype=float)
    m[m < np.amax(m) * 1e-20] = 0                       # tiny values create image artefacts after the gamma correction
    marginal_image(stamps, bounds, m, color=color, gamma=gamma)
    if 'c' not in kwargs and 'color' not in kwargs:
        kwargs['c'] = 'k'
    if 'lw' not in kwargs and 'linewidth' not in kwargs:
        kwargs['lw'] = 1.5
    plt.plot(stamps, means, **kwargs)
    plt.ylim(bounds)
    plt.ylabel(ylabel)
    plt.xlabel('time step')


def bars(x, p, xlabel, width=None, **kwargs):
    """Discrete distribution over hyper-parameter values: categorical axis if the values are not equally spaced."""
    plt = _plt()
    x = np.asarray(x, dtype=float)
    if _is_regular(x):
        if width is None:
            width = (x[1] - x[0]) if len(x) > 1 else 1.0
        plt.bar(x, p, align='center', width=width, **kwargs)
    else:
        plt.bar(np.arange(len(x)), p, align='center', width=1.0, **kwargs)
        plt.xticks(np.arange(len(x)), x)
    plt.ylabel('probability')
    plt.xlabel(xlabel)


def joint_bars(x, y, z, names, widths, figure=None, subplot=111, **kwargs):
    plt = _plt()
    from mpl_toolkits.mplot3d import Axes3D  # noqa: F401  (registers the projection on old matplotlib)
    fig = plt.figure() if figure is None else figure
    ax = fig.add_subplot(subplot, projection='3d')
    X, Y = np.meshgrid(x, y, indexing='ij')
    Z = np.asarray(z, dtype=float)
    ax.bar3d(X.ravel() - widths[0] / 2.0, Y.ravel() - widths[1] / 2.0, np.zeros(Z.size), widths[0], widths[1], Z.ravel(),
             zsort='max', **kwargs)
    ax.set_xlabel(names[0])
    ax.set_ylabel(names[1])
    ax.set_zlabel('probability')
    return ax


def line(stamps, values, **kwargs):
    _plt().plot(stamps, values, **kwargs)
