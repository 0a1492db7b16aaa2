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
