"""Data formatting (API parity with bayesloop/preprocessing.py:14-26 of the reference)."""
import numpy as np


def movingWindow(rawData, n):
    """Overlapping data segments of length ``n``: shape (len(rawData) - n + 1, n[, d])."""
    rawData = np.asarray(rawData)
    count = rawData.shape[0] - (n - 1)
    return np.array([rawData[k:k + n] for k in range(count)])
