"""ctypes wrappers of the two CPU lattices (TEST INFRASTRUCTURE):

* ``libpermuto_oracle.so``       - our plain-C restatement (oracle/permutohedral_oracle.c)
* ``_ref/libpermuto_ref.so``     - the reference's vendored permutohedral.cpp compiled verbatim
                                   (oracle/Makefile target ``ref``; exists only if it was built where
                                   /root/reference is present - it then travels with the snapshot)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpermuto_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libpermuto_ref.so")
_lib = None
_ref = None


def _oracle():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "permutohedral_oracle.c")
        if not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libpermuto_oracle.so"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(_SO)
        L.permuto_oracle_create.restype = ctypes.c_void_p
        L.permuto_oracle_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.permuto_oracle_size.argtypes = [ctypes.c_void_p]
        L.permuto_oracle_get.argtypes = [ctypes.c_void_p] * 4
        L.permuto_oracle_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.permuto_oracle_destroy.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def ref_available():
    return os.path.isfile(_REF_SO)


def _reflib():
    global _ref
    if _ref is None:
        L = ctypes.CDLL(_REF_SO)
        L.permuto_ref_create.restype = ctypes.c_void_p
        L.permuto_ref_destroy.argtypes = [ctypes.c_void_p]
        L.permuto_ref_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.permuto_ref_lattice_size.argtypes = [ctypes.c_void_p]
        L.permuto_ref_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _ref = L
    return _ref


class Lattice(object):
    """Permutohedral lattice over ``points`` (n x d).  ``prefer_ref`` uses the vendored reference build."""

    def __init__(self, points, with_blur=True, prefer_ref=False):
        self.points = np.ascontiguousarray(points, dtype=np.float32)
        self.n, self.d = self.points.shape
        self.with_blur = bool(with_blur)
        self.is_ref = bool(prefer_ref and ref_available())
        if self.is_ref:
            self._h = _reflib().permuto_ref_create()
            # n x d row-major == d x n column-major, the layout Permutohedral::init reads
            _reflib().permuto_ref_init(self._h, self.points.ctypes.data, self.d, self.n, int(self.with_blur))
            self.lattice_size = int(_reflib().permuto_ref_lattice_size(self._h))
        else:
            self._h = _oracle().permuto_oracle_create(self.points.ctypes.data, self.n, self.d, int(self.with_blur))
            if not self._h:
                raise ValueError("unsupported lattice dimension %d" % self.d)
            self.lattice_size = int(_oracle().permuto_oracle_size(self._h))

    def structure(self):
        """(offset [n, d+1] int32, barycentric [n, d+1] float32, keys [m, d] int16) - oracle build only."""
        assert not self.is_ref
        off = np.empty((self.n, self.d + 1), dtype=np.int32)
        bar = np.empty((self.n, self.d + 1), dtype=np.float32)
        keys = np.empty((self.lattice_size, self.d), dtype=np.int16)
        _oracle().permuto_oracle_get(self._h, off.ctypes.data, bar.ctypes.data, keys.ctypes.data)
        return off, bar, keys

    def filter(self, values):
        """values n x ch -> filtered n x ch (float32)."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        if v.ndim == 1:
            v = v[:, None]
        out = np.empty_like(v)
        if self.is_ref:
            _reflib().permuto_ref_filter(self._h, v.ctypes.data, v.shape[1], self.n, out.ctypes.data)
        else:
            _oracle().permuto_oracle_filter(self._h, v.ctypes.data, v.shape[1], out.ctypes.data)
        return out

    def __del__(self):
        try:
            if self._h:
                if self.is_ref:
                    _reflib().permuto_ref_destroy(self._h)
                else:
                    _oracle().permuto_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass
