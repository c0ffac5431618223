"""Direct Gauss transform on the GPU (reference probreg/gauss_transform.py:10-60).

The reference switches to an IFGT approximation for ``h >= sw_h``; on MI355X the exact O(S*T)
sum is a streaming kernel of the same shape as the CPD column pass, so ``GaussTransform`` is
always exact here (``eps`` / ``sw_h`` are accepted for signature compatibility).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib, ptr
from .engine import _current_device_and_stream


def _gauss_transform_direct(source, target, weights, h):
    r"""\sum_j weights[j] * exp(-||target[i] - source[j]||^2 / h^2)   (gauss_transform.py:10-16)."""
    _lib.require_gpu()
    source = np.ascontiguousarray(source, dtype=np.float64)  # float64 across the ABI: differences are formed in fp64
    target = np.ascontiguousarray(target, dtype=np.float64)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    if weights.ndim == 1:
        rows = 1
        wmat = weights[None, :]
    elif weights.ndim == 2:
        rows = weights.shape[0]
        wmat = weights
    else:
        raise ValueError("weights.ndim must be 1 or 2.")
    if wmat.shape[1] != source.shape[0]:
        raise ValueError("weights must have one entry per source point.")
    dev, st = _current_device_and_stream()
    out = np.empty((rows, target.shape[0]), dtype=np.float64)
    check(lib.prg_gauss_transform_direct(dev, ctypes.c_void_p(st), ptr(source), source.shape[0], ptr(target),
                                         target.shape[0], source.shape[1], ptr(np.ascontiguousarray(wmat)), rows,
                                         float(h), ptr(out)))
    return out[0] if weights.ndim == 1 else out


class Direct(object):
    def __init__(self, source, h):
        self._source = source
        self._h = h

    def compute(self, target, weights):
        return _gauss_transform_direct(self._source, target, weights, self._h)


class GaussTransform(object):
    """``GaussTransform(source, h).compute(target, weights)`` as in reference gauss_transform.py:28-60: weights may be
    one row (S) or several (C x S); ``eps`` and ``sw_h`` only mattered for the reference's IFGT switch."""

    def __init__(self, source, h, eps=1.0e-4, sw_h=0.01):
        self._m = source.shape[0]
        self._impl = Direct(source, h)

    def compute(self, target, weights=None):
        if weights is None:
            weights = np.ones(self._m)
        weights = np.asarray(weights)
        if weights.ndim not in (1, 2):
            raise ValueError("weights.ndim must be 1 or 2.")
        return self._impl.compute(target, weights)
