"""``probreg.math_utils`` counterparts that are on the hot path (reference probreg/math_utils.py:28-37)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib, ptr
from .engine import _current_device_and_stream


def squared_kernel_sum(x, y):
    """sum_{m,n} |x_m - y_n|^2 / (M*D*N)  (reference math_utils.py:28-29 -> cc/math_utils.cc:5-15).

    Evaluated on the GPU in closed form in fp64 - the M x N float32 matrix of the reference is
    never built; the two agree to float32 rounding of the reference's sum (~1e-7 relative).
    """
    _lib.require_gpu()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    if x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1]:
        raise ValueError("x and y must be 2-D with the same number of columns.")
    dev, st = _current_device_and_stream()
    out = ctypes.c_double(0.0)
    check(lib.prg_squared_kernel_sum(dev, ctypes.c_void_p(st), ptr(x), x.shape[0], ptr(y), y.shape[0], x.shape[1],
                                     ctypes.byref(out)))
    return float(out.value)


def rbf_kernel(x, y, beta):
    """K_ij = exp(-|x_i - y_j|^2 / (2*beta)) as float32 (reference math_utils.py:36-37 -> cc/math_utils.cc:17-19)."""
    _lib.require_gpu()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    if x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1]:
        raise ValueError("x and y must be 2-D with the same number of columns.")
    dev, st = _current_device_and_stream()
    out = np.empty((x.shape[0], y.shape[0]), dtype=np.float32)
    check(lib.prg_rbf_kernel(dev, ctypes.c_void_p(st), ptr(x), x.shape[0], ptr(y), y.shape[0], x.shape[1],
                             float(beta), ptr(out)))
    return out
