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


def inverse_multiquadric_kernel(x, y, c=1.0):
    """K_ij = 1 / sqrt(|x_i - y_j|^2 + c) as float32 (reference math_utils.py:50-51 -> cc/math_utils.cc:32-34)."""
    _lib.require_gpu()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    if x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1]:
        raise ValueError("x and y must be 2-D with the same number of columns.")
    dev, st = _current_device_and_stream()
    out = np.empty((x.shape[0], y.shape[0]), dtype=np.float32)
    check(lib.prg_inverse_multiquadric_kernel(dev, ctypes.c_void_p(st), ptr(x), x.shape[0], ptr(y), y.shape[0],
                                              x.shape[1], float(c), ptr(out)))
    return out


def compute_rmse(source, target):
    """Mean distance from every source point to its nearest target point (reference math_utils.py:32-33).

    The reference takes a prebuilt ``cKDTree`` of the target; here ``target`` is the point array itself (an object
    with a ``.data`` attribute, such as a cKDTree, is accepted too) and the search is a brute-force GPU sweep.
    Both clouds are shifted by the target centroid in float64 before the float32 upload.
    """
    _lib.require_gpu()
    tgt = np.asarray(getattr(target, "data", target), dtype=np.float64)
    src = np.asarray(source, dtype=np.float64)
    if src.ndim != 2 or tgt.ndim != 2 or src.shape[1] != tgt.shape[1]:
        raise ValueError("source and target must be 2-D with the same number of columns.")
    c = tgt.mean(axis=0)
    a = np.ascontiguousarray(src - c, dtype=np.float32)
    b = np.ascontiguousarray(tgt - c, dtype=np.float32)
    dev, st = _current_device_and_stream()
    out = ctypes.c_double(0.0)
    check(lib.prg_nn_mean_distance(dev, ctypes.c_void_p(st), ptr(a), a.shape[0], ptr(b), b.shape[0], a.shape[1],
                                   ctypes.byref(out)))
    return float(out.value)
