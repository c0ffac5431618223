"""Load the UNMODIFIED reference modules from /root/reference (build container only).

TEST INFRASTRUCTURE.  The reference package cannot be imported as shipped here:
``open3d``, ``transforms3d``, ``cupy``, ``dq3d`` are not installed and its six
pybind11 extensions need Eigen, which is an empty submodule.  This module puts
small stand-ins into ``sys.modules`` for exactly those names and then lets
``importlib`` execute the reference's own ``probreg/cpd.py``,
``probreg/transformation.py``, ``probreg/math_utils.py`` and
``probreg/filterreg.py`` as they lie on disk.

Stand-ins and what pins them:
  ``open3d``                  only used for ``isinstance`` (cpd.py:444, transformation.py:23-24)
  ``probreg._math``           fp32 restatement of cc/math_utils.cc:5-19, 32-34 (kernelBase, squaredKernel,
                              rbfKernel - note ``2*beta`` not ``2*beta**2`` -, inverseMultiQuadricKernel); pinned by the
                              reference test tests/test_math_utils.py:7-16
  ``probreg._kabsch``         fp32 restatement of cc/kabsch.cc:6-109 (weights squared in H,
                              unsquared in the centroids)
  ``probreg._permutohedral_lattice``  the VENDORED third_party/permutohedral/permutohedral.cpp
                              compiled verbatim into oracle/_ref (see oracle/Makefile) behind a
                              tiny Eigen shim; falls back to the C restatement when _ref is absent
  ``probreg._pt2pl`` / ``_ifgt`` / ``transforms3d``  dummies (not touched on the pt2pt path)

Nothing here is available on the GPU box (no /root/reference there): only
``tests/golden/make_golden.py`` and the container-side oracle self-checks use it.
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("PROBREG_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "probreg", "cpd.py"))


def _math_standin():
    m = types.ModuleType("probreg._math")

    def _sqdist_f32(x, y):
        # cc/math_utils.cc:5-13: K(:, i) = || x_r - y_i ||^2 evaluated in float32.
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        k = np.empty((x.shape[0], y.shape[0]), dtype=np.float32)
        for i in range(y.shape[0]):
            d = x - y[i]
            k[:, i] = np.einsum("rd,rd->r", d, d, dtype=np.float32)
        return k

    def squared_kernel(x, y):
        return _sqdist_f32(x, y)

    def rbf_kernel(x, y, beta):
        # cc/math_utils.cc:17-19: exp(-d2 / (2.0*beta)); Eigen evaluates the float vector
        # divided by a double scalar in float (scalar is cast to the vector's scalar type).
        d2 = _sqdist_f32(x, y)
        return np.exp(-d2 / np.float32(2.0 * beta)).astype(np.float32)

    def inverse_multiquadric_kernel(x, y, c):
        # cc/math_utils.cc:32-34: 1.0 / sqrt(d2 + c) on the float32 array (the scalars are cast to float)
        d2 = _sqdist_f32(x, y)
        return (np.float32(1.0) / np.sqrt(d2 + np.float32(c))).astype(np.float32)

    m.squared_kernel = squared_kernel
    m.rbf_kernel = rbf_kernel
    m.inverse_multiquadric_kernel = inverse_multiquadric_kernel
    m.tps_kernel_2d = m.tps_kernel_3d = None
    return m


def _kabsch_standin():
    from . import filterreg_numpy as fo

    m = types.ModuleType("probreg._kabsch")
    m.kabsch = fo.kabsch_f32
    m.kabsch2d = fo.kabsch2d_f32
    return m


def _permutohedral_standin():
    from . import permutohedral as ph

    m = types.ModuleType("probreg._permutohedral_lattice")

    class Permutohedral(object):
        def __init__(self):
            self._impl = None

        def init(self, features, with_blur):
            # features: d x N (the python shim passes p.T, gaussian_filtering.py:11)
            self._impl = ph.Lattice(np.asarray(features, dtype=np.float32).T, with_blur, prefer_ref=True)

        def get_lattice_size(self):
            return self._impl.lattice_size

        def filter(self, v, start):
            # v: channels x N ; returns channels x N (permutohedral_lattice_py.cc:16-20)
            return self._impl.filter(np.asarray(v, dtype=np.float32).T).T

    m.Permutohedral = Permutohedral
    return m


_loaded = {}


def load_bcpd():
    """The reference's ``probreg.bcpd`` module (needs scipy, six - both installed - and the _math stand-in)."""
    load(with_filterreg=False)
    return importlib.import_module("probreg.bcpd")


def load_gauss():
    """The reference's ``probreg.gauss_transform`` and ``probreg.cost_functions`` as they lie on disk.

    ``probreg._ifgt`` (cc/ifgt.cc needs Eigen: not buildable here) is stood in for by the reference's OWN ``Direct``
    class (gauss_transform.py:19-25): IFGT approximates exactly that sum to ``eps`` (1e-4 by default), so above the
    ``sw_h`` switch the fixtures hold the value the reference's IFGT approximates, below it the value its direct path returns.
    ``transforms3d`` is only touched by the rigid cost function's quaternion code, never by ``compute_l2_dist``."""
    if "gauss" in _loaded:
        return _loaded["gauss"]
    load(with_filterreg=False)
    for name in ("transforms3d", "transforms3d.quaternions", "transforms3d.euler"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "probreg._ifgt" not in sys.modules or getattr(sys.modules["probreg._ifgt"], "Ifgt", None) is None:
        sys.modules["probreg._ifgt"] = types.ModuleType("probreg._ifgt")
    sys.modules["probreg._ifgt"].Ifgt = None
    ns = types.SimpleNamespace()
    ns.gauss_transform = importlib.import_module("probreg.gauss_transform")

    class _IfgtAsDirect(object):  # same constructor / compute signature as cc/ifgt_py.cc:14
        def __init__(self, source, h, eps):
            self._impl = ns.gauss_transform.Direct(np.asarray(source, dtype=np.float64), h)

        def compute(self, target, weights):
            return self._impl.compute(np.asarray(target, dtype=np.float64), np.asarray(weights, dtype=np.float64))

    sys.modules["probreg._ifgt"].Ifgt = _IfgtAsDirect
    ns.gauss_transform._ifgt.Ifgt = _IfgtAsDirect
    ns.cost_functions = importlib.import_module("probreg.cost_functions")
    _loaded["gauss"] = ns
    return ns


def load(with_filterreg=False):
    """Return a namespace with the reference's ``cpd`` (and optionally ``filterreg``) modules."""
    if not available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    key = bool(with_filterreg)
    if key in _loaded:
        return _loaded[key]

    if "open3d" not in sys.modules:
        o3 = types.ModuleType("open3d")
        o3.geometry = types.ModuleType("open3d.geometry")
        o3.utility = types.ModuleType("open3d.utility")

        class PointCloud(object):
            pass

        class Vector3dVector(object):
            pass

        o3.geometry.PointCloud = PointCloud
        o3.utility.Vector3dVector = Vector3dVector
        sys.modules["open3d"] = o3
        sys.modules["open3d.geometry"] = o3.geometry
        sys.modules["open3d.utility"] = o3.utility

    if "probreg" not in sys.modules or not getattr(sys.modules["probreg"], "_oracle_stub", False):
        pkg = types.ModuleType("probreg")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "probreg")]
        pkg._oracle_stub = True
        sys.modules["probreg"] = pkg
        sys.modules["probreg._math"] = _math_standin()

    ns = types.SimpleNamespace()
    ns.cpd = importlib.import_module("probreg.cpd")
    ns.transformation = importlib.import_module("probreg.transformation")
    ns.math_utils = importlib.import_module("probreg.math_utils")

    if with_filterreg:
        if "transforms3d" not in sys.modules:
            t3 = types.ModuleType("transforms3d")
            t3.quaternions = types.ModuleType("transforms3d.quaternions")
            t3.euler = types.ModuleType("transforms3d.euler")
            sys.modules["transforms3d"] = t3
            sys.modules["transforms3d.quaternions"] = t3.quaternions
            sys.modules["transforms3d.euler"] = t3.euler
        sys.modules.setdefault("probreg._kabsch", _kabsch_standin())
        pt2pl_mod = types.ModuleType("probreg._pt2pl")
        from . import filterreg_numpy as _fo

        pt2pl_mod.compute_twist_for_pt2pl = _fo.pt2pl_f32  # float32 restatement of cc/point_to_plane.cc:6-32
        sys.modules.setdefault("probreg._pt2pl", pt2pl_mod)
        ifgt = types.ModuleType("probreg._ifgt")
        ifgt.Ifgt = None
        sys.modules.setdefault("probreg._ifgt", ifgt)
        sys.modules.setdefault("probreg._permutohedral_lattice", _permutohedral_standin())
        ns.filterreg = importlib.import_module("probreg.filterreg")
        ns.gauss_transform = importlib.import_module("probreg.gauss_transform")
    _loaded[key] = ns
    return ns
