"""Permutohedral-lattice Gaussian filtering on the GPU (reference probreg/gaussian_filtering.py:8-17)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib, ptr
from .engine import _current_device_and_stream


class Permutohedral(object):
    """``Permutohedral(p, with_blur)`` / ``get_lattice_size()`` / ``filter(v, start)`` as in the reference.

    ``p`` is (n, d), 1 <= d <= 64: d <= 3 (positions) takes the packed-key kernels, larger d (feature lattices such as
    the reference's 33-dimensional FPFH, features.py:28-51) the hashed-key ones.
    """

    def __init__(self, p, with_blur=True):
        _lib.require_gpu()
        p = np.ascontiguousarray(p, dtype=np.float32)
        if p.ndim != 2:
            raise ValueError("p must be a 2-D array (points x features).")
        dev, st = _current_device_and_stream()
        self._h = ctypes.c_void_p()
        check(lib.prg_ph_create(ctypes.byref(self._h), dev, ctypes.c_void_p(st)))
        self._n = p.shape[0]
        check(lib.prg_ph_init(self._h, ptr(p), p.shape[0], p.shape[1], 1 if with_blur else 0))

    def get_lattice_size(self):
        n = ctypes.c_int(0)
        check(lib.prg_ph_lattice_size(self._h, ctypes.byref(n)))
        return int(n.value)

    def filter(self, v, start=0):
        # ``start`` is accepted and ignored exactly like the reference's compute() ignores it
        v = np.ascontiguousarray(v, dtype=np.float32)
        if v.ndim == 1:
            v = v[:, None]
        if v.shape[0] != self._n:
            raise ValueError("v must have one row per lattice point.")
        out = np.empty_like(v)
        check(lib.prg_ph_filter(self._h, ptr(v), v.shape[1], ptr(out)))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.prg_ph_destroy(self._h)
                self._h = None
        except Exception:  # pragma: no cover
            pass
