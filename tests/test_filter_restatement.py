"""The restated scipy.ndimage.gaussian_filter1d (oracle) against SciPy itself, where SciPy is installed."""
import numpy as np
import pytest

from oracle import bl_oracle as orc

scipy_ndimage = pytest.importorskip('scipy.ndimage')


@pytest.mark.parametrize('shape,axis,sigma', [
    ((200,), 0, 6.7), ((4096,), 0, 5.119), ((64, 48), 0, 1.918), ((64, 48), 1, 2.05), ((40,), 0, 23.3),
    ((7, 5), 0, 3.0), ((7, 5), 1, 9.0), ((33,), 0, 0.13), ((33,), 0, 0.124), ((3, 3), 1, 0.6)])
def test_bit_exact_against_scipy(shape, axis, sigma):
    rng = np.random.default_rng(abs(hash((shape, axis))) % 2 ** 31)
    x = rng.random(shape)
    want = scipy_ndimage.gaussian_filter1d(x, sigma, axis=axis)
    got = orc.gaussian_filter1d(x, sigma, axis)
    assert np.array_equal(want, got)


def test_kernel_radius_and_identity():
    lw, w = orc.gaussian_kernel1d(0.124)
    assert lw == 0 and w.tolist() == [1.0]
    lw, w = orc.gaussian_kernel1d(6.7)
    assert lw == 27 and len(w) == 55 and abs(w.sum() - 1) < 1e-15


@pytest.mark.parametrize('n', [5, 20, 57, 100])
@pytest.mark.parametrize('d', [0.0, 0.4, -0.7, 3.3, 6.73, -6.73, 11.9, 12.0, 12.3, 13.37, -25.5, 31.2, -60.0])
def test_spline_shift_against_scipy(n, d):
    """Deterministic model (reference transitionModels.py:581, :600): the oracle's restatement of
    scipy.ndimage.shift(order=3, mode='nearest') -- 12-sample edge pad, reflect-mode prefilter, edge-extended coefficients."""
    x = np.random.default_rng(n * 1000 + int(abs(d) * 10)).random(n) ** 3
    want = scipy_ndimage.shift(x, d, order=3, mode='nearest')
    got = orc.spline_shift_nearest(x, d, 0)
    assert np.max(np.abs(want - got)) < 5e-15


def test_spline_shift_against_scipy_2d():
    x = np.random.default_rng(5).random((20, 31))
    assert np.max(np.abs(scipy_ndimage.shift(x, [0, 2.2], order=3, mode='nearest') - orc.spline_shift_nearest(x, 2.2, 1))) < 5e-15
    assert np.max(np.abs(scipy_ndimage.shift(x, [-13.7, 0], order=3, mode='nearest') - orc.spline_shift_nearest(x, -13.7, 0))) < 5e-15


def test_bivariate_kernel_and_convolution_against_scipy():
    """BivariateRandomWalk (reference transitionModels.py:889, :898-911)."""
    stats = pytest.importorskip('scipy.stats')
    signal = pytest.importorskip('scipy.signal')
    s1, s2, rho = 3.5, 1.05, 0.5
    rv = stats.multivariate_normal(cov=[[s1 ** 2., rho * s1 * s2], [rho * s1 * s2, s2 ** 2.]])
    xs = np.arange(-3 * np.ceil(s1), 3 * np.ceil(s1) + 1)
    ys = np.arange(-3 * np.ceil(s2), 3 * np.ceil(s2) + 1)
    xv, yv = np.meshgrid(xs, ys, indexing='ij')
    want = rv.pdf(np.array([xv, yv]).T).T
    want /= np.sum(want)
    got = orc.bivariate_kernel(s1, s2, rho)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    x = np.random.default_rng(6).random((20, 17))
    np.testing.assert_allclose(orc.convolve2d_same_zero(x, got), signal.convolve2d(x, got, mode='same'), rtol=1e-13, atol=1e-16)


def test_alphastable_convolution_against_scipy():
    """AlphaStableRandomWalk (reference transitionModels.py:196-260): kernel roll + 3x padding + fftconvolve(mode='same')
    against the oracle's direct symmetric Toeplitz sum, even and odd grid sizes."""
    signal = pytest.importorskip('scipy.signal')
    for n in (100, 37):
        c, alpha = 3.366, 1.5
        kernel_fft = np.exp(-np.abs(c * np.linspace(0, np.pi, int(3 * n / 2 + 1))) ** alpha)
        kernel = np.roll(np.fft.irfft(kernel_fft), int(3 * n / 2 - 1))
        x = np.random.default_rng(n).random(n)
        padded = np.zeros(3 * n)
        padded[n:2 * n] = x
        want = signal.fftconvolve(padded, kernel, mode='same')[n:2 * n]
        got = orc.convolve_axis_zero(x, orc.alphastable_kernel(c, alpha, n), 0)
        assert np.max(np.abs(want - got)) < 1e-13


@pytest.mark.parametrize('n', [2, 3, 5, 24, 100, 224, 1024, 4120])
def test_spline_prefilter_recursion_is_scipys_bit_for_bit(n):
    """oracle/spline_iir.c (and its pure-Python twin) = scipy.ndimage.spline_filter1d(x, 3, mode='nearest') to the last bit: SciPy's
    recursion (ni_splines.c: gain, _init_causal_reflect, causal pass, _init_anticausal_reflect, anti-causal pass) with SciPy's pole
    literal -0.2679491924311227 -- rows of every magnitude, incl. tails that decay into the denormals."""
    rng = np.random.default_rng(n)
    x = np.arange(n)
    rows = np.stack([rng.random(n), rng.random(n) * np.exp(rng.normal(0, 40, n)), np.exp(-0.5 * ((x - 0.6 * n) / 3.0) ** 2),
                     np.exp(-x / 2.0), 1e-300 * rng.random(n), np.where(x == n // 2, 1.0, 0.0)])
    want = scipy_ndimage.spline_filter1d(rows, 3, axis=1, mode='nearest')
    got = orc.spline_prefilter_reflect(rows.copy())
    assert np.array_equal(got, want)
    # the fall-back without a C compiler: the same recursion in numpy, the same bits
    lib, orc._SPLINE_LIB[:] = list(orc._SPLINE_LIB), [None]
    try:
        assert np.array_equal(orc.spline_prefilter_reflect(rows.copy()), want)
    finally:
        orc._SPLINE_LIB[:] = lib


def test_spline_shift_off_the_grid_keeps_the_recursions_tail():
    """A distribution shifted off the grid: what is left is the prefilter's tail (here ~1e-130 of the mass), alternating in sign --
    the restatement agrees with SciPy in every cell to rounding RELATIVE TO THE CELL; the truncated response of rounds 1 - 4 left
    exact zeros there."""
    n = 200
    x = np.exp(-0.5 * ((np.arange(n) - 150.0) / 1.5) ** 2)
    x /= x.sum()
    for d in (-260.3, 255.0):
        want = scipy_ndimage.shift(x, d, order=3, mode='nearest')
        got = orc.spline_shift_nearest(x, d, 0)
        assert np.max(np.abs(want)) < 1e-20 and np.all(want != 0.0)
        assert np.all(np.abs(got - want) <= 1e-12 * np.abs(want))
