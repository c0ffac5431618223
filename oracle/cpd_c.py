"""ctypes wrapper of the C E-step restatement (oracle/cpd_estep_c.c).  TEST INFRASTRUCTURE / CPU baseline."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcpd_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "cpd_estep_c.c")):
        subprocess.check_call(["make", "-C", _HERE, "libcpd_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.cpd_oracle_estep.restype = ctypes.c_double
        _lib.cpd_oracle_estep.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p]
        _lib.cpd_oracle_threads.restype = ctypes.c_int
    return _lib


def threads():
    return int(lib().cpd_oracle_threads())


def expectation_step(t_source, target, sigma2, w=0.0):
    """cpd.py:71-88 in C/OpenMP float64; returns (pt1, p1, px, n_p) like oracle.cpd_numpy.expectation_step."""
    ts = np.ascontiguousarray(t_source, dtype=np.float64)
    x = np.ascontiguousarray(target, dtype=np.float64)
    m, d = ts.shape
    n = x.shape[0]
    pt1 = np.empty(n)
    p1 = np.empty(m)
    px = np.empty((m, d))
    n_p = lib().cpd_oracle_estep(ts.ctypes.data, m, x.ctypes.data, n, d, float(sigma2), float(w), pt1.ctypes.data,
                                 p1.ctypes.data, px.ctypes.data)
    return pt1, p1, px, float(n_p)


def nonrigid_lhs(source, beta, p1, c):
    """``(p1 * g).T + c I`` of cpd.py:297 as a column-major (LAPACK-ready) float64 matrix, g = the float32 kernel matrix
    of ``source`` evaluated on the fly (oracle.cpd_numpy.rbf_kernel entry by entry)."""
    y = np.ascontiguousarray(source, dtype=np.float32)
    m, d = y.shape
    p1 = np.ascontiguousarray(p1, dtype=np.float64)
    a = np.empty((m, m), dtype=np.float64, order="F")
    fn = lib().cpd_oracle_nonrigid_lhs
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
                   ctypes.c_void_p]
    fn(y.ctypes.data, m, d, float(beta), p1.ctypes.data, float(c), a.ctypes.data)
    return a


def nonrigid_gw(source, beta, w):
    """``g @ w`` (cpd.py:298) with the float32 kernel matrix evaluated on the fly, float64 accumulation."""
    y = np.ascontiguousarray(source, dtype=np.float32)
    m, d = y.shape
    w = np.ascontiguousarray(w, dtype=np.float64)
    assert w.shape == (m, d) and d <= 3
    out = np.empty((m, d), dtype=np.float64)
    fn = lib().cpd_oracle_nonrigid_gw
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    fn(y.ctypes.data, m, d, float(beta), w.ctypes.data, out.ctypes.data)
    return out
