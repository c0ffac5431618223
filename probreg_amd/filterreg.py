"""FilterReg (rigid, point-to-point) on MI355X - drop-in for ``probreg.filterreg`` on that path
(reference probreg/filterreg.py:27-196, 269-317).

The E-step (lattice build over [t_source; target]/sigma, the three Gaussian filters) and the M-step
(weighted Kabsch, composition, sigma2 update) run in ``libprobreg_hip.so``; the Python loop mirrors
``FilterReg.registration`` (filterreg.py:120-147) line by line, including its quirks: with the defaults
sigma2 is never updated, the returned ``sigma2`` is the un-clamped one, and the driver stops with the
previous ``q`` when every ``m0`` is zero.
Point-to-plane (``objective_type='pt2pl'``, filterreg.py:183-186 + cc/point_to_plane.cc) is built as well, and so are
feature-space lattices: a ``feature_fn`` other than the identity (filterreg.py:121, 125-133; e.g. 33-dimensional FPFH
descriptors) runs the reference's loop - transform, ``feature_fn`` on the host, lattice E-step over the features
(1 <= d <= 64) and M-step on the GPU.  Out of scope: ``DeformableKinematicFilterReg``.
"""
import abc
import ctypes
from collections import namedtuple

import numpy as np

from . import _lib
from . import math_utils as mu
from . import transformation as tf
from ._lib import check, lib, ptr
from .engine import _current_device_and_stream
from .log import log

EstepResult = namedtuple("EstepResult", ["m0", "m1", "m2", "nx"])
MstepResult = namedtuple("MstepResult", ["transformation", "sigma2", "q"])
MstepResult.__doc__ = """(transformation, sigma2, q) of one FilterReg M-step / of the whole registration (reference filterreg.py:28)."""


def _as_points(x):
    if x is None:
        return None
    if hasattr(x, "points") and not isinstance(x, np.ndarray):
        x = x.points
    return np.asarray(x, dtype=np.float64)


def _identity(x):
    return x


class _Plan(object):
    """One ``prg_filterreg`` handle."""

    def __init__(self, device=None):
        _lib.require_gpu()
        dev, st = _current_device_and_stream(device)
        self._h = ctypes.c_void_p()
        check(lib.prg_fr_create(ctypes.byref(self._h), dev, ctypes.c_void_p(st)))
        self.m = self.n = self.dim = 0

    def set_source(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.m, self.dim = a.shape
        check(lib.prg_fr_set_source(self._h, ptr(a), a.shape[0], a.shape[1]))

    def set_target(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.n = a.shape[0]
        check(lib.prg_fr_set_target(self._h, ptr(a), a.shape[0], a.shape[1]))

    def set_state(self, rot, t, sigma2):
        r = np.identity(3)
        d = np.asarray(rot).shape[0]
        r[:d, :d] = rot
        tt = np.zeros(3)
        tt[:d] = t
        check(lib.prg_fr_set_state(self._h, ptr(np.ascontiguousarray(r)), ptr(tt), float(sigma2)))

    def estep(self, alpha=0.015):
        size, blur = ctypes.c_int(0), ctypes.c_int(0)
        check(lib.prg_fr_estep(self._h, float(alpha), ctypes.byref(size), ctypes.byref(blur)))
        return int(size.value), bool(blur.value)

    def get_estep(self, want_m2):
        m0 = np.empty(self.m, dtype=np.float32)
        m1 = np.empty((self.m, self.dim), dtype=np.float32)
        m2 = np.empty(self.m, dtype=np.float32) if want_m2 else None
        check(lib.prg_fr_get_estep(self._h, ptr(m0), ptr(m1), ptr(m2) if want_m2 else None))
        return m0, m1, m2

    def set_normals(self, normals):
        if normals is None:
            check(lib.prg_fr_set_target_normals(self._h, None))
        else:
            a = np.ascontiguousarray(normals, dtype=np.float64)
            if a.shape != (self.n, 3):
                raise ValueError("target_normals must be an (n, 3) array matching the target.")
            check(lib.prg_fr_set_target_normals(self._h, ptr(a)))

    def get_nx(self):
        nx = np.empty((self.m, 3), dtype=np.float32)
        check(lib.prg_fr_get_nx(self._h, ptr(nx)))
        return nx

    def mstep(self, w, update_sigma2, objective_type="pt2pt", min_sigma2=-1.0, read=True):
        """``min_sigma2`` >= 0 advances the device sigma2 like the driver (filterreg.py:140); negative leaves it alone.
        ``read=False`` only enqueues the M-step (no read-back, the host does not wait) and returns None."""
        fn = lib.prg_fr_mstep_pt2pl if objective_type == "pt2pl" else lib.prg_fr_mstep
        if not read:
            check(fn(self._h, float(w), 1 if update_sigma2 else 0, float(min_sigma2), None))
            return None
        out = np.zeros(18)
        check(fn(self._h, float(w), 1 if update_sigma2 else 0, float(min_sigma2), ptr(out)))
        return out

    def get_state(self):
        """The 20-double device state (prg_fr_get_state); synchronises."""
        out = np.zeros(20)
        check(lib.prg_fr_get_state(self._h, ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib.prg_fr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


class FilterReg(abc.ABC):
    """Gaussian-filter based EM registration, driver part (reference filterreg.py:45-147).

    source         : (M, D) array that gets moved
    target_normals : (N, 3) unit normals of the target, only for ``objective_type='pt2pl'``
    sigma2         : kernel variance; ``None`` = mean squared source-target distance (clamped to ``min_sigma2``)
    update_sigma2  : re-estimate sigma2 in every M-step (off by default, as in the reference)
    """

    def __init__(self, source=None, target_normals=None, sigma2=None, update_sigma2=False):
        self._source = _as_points(source)
        self._target_normals = target_normals
        self._sigma2 = sigma2
        self._update_sigma2 = update_sigma2
        self._tf_type = None
        self._tf_result = None
        self._callbacks = []
        self._plan = None

    def set_source(self, source):
        self._source = _as_points(source)

    def set_target_normals(self, target_normals):
        self._target_normals = target_normals

    def set_callbacks(self, callbacks):
        self._callbacks = callbacks

    def _ensure_plan(self, target):
        if self._plan is None:
            self._plan = _Plan()
        self._plan.set_source(self._source)
        self._plan.set_target(target)
        return self._plan

    def expectation_step(self, t_source, target, y, sigma2, update_sigma2, objective_type="pt2pt", alpha=0.015):
        """Expectation step (reference filterreg.py:78-108) on explicit arrays; returns float32 m0, m1, m2 (, nx).

        ``t_source`` / ``target`` are the FEATURES the lattice is built over (the transformed source and the target
        themselves for the identity ``feature_fn``, anything up to 64-dimensional otherwise) and ``y`` the target
        positions whose moments are filtered.
        """
        t_source, target, y = np.asarray(t_source), np.asarray(target), np.asarray(y)
        assert t_source.ndim == 2 and target.ndim == 2, "source and target must have 2 dimensions."
        if objective_type not in ("pt2pt", "pt2pl"):
            raise ValueError("Unknown objective_type: %s." % objective_type)
        if t_source.shape[1] <= 3 and y.shape == target.shape and np.array_equal(y, target):
            # positions as features: the fused plan (one lattice pass for all channels)
            plan = _Plan()
            try:
                plan.set_source(t_source)
                plan.set_target(target)
                if objective_type == "pt2pl":
                    plan.set_normals(self._target_normals)
                plan.set_state(np.identity(t_source.shape[1]), np.zeros(t_source.shape[1]), sigma2)
                plan.estep(alpha)
                m0, m1, m2 = plan.get_estep(update_sigma2)
                nx = plan.get_nx() if objective_type == "pt2pl" else None
            finally:
                plan.close()
            return EstepResult(m0, m1, m2, nx)
        return self._expectation_step_features(t_source, target, y, sigma2, update_sigma2, objective_type, alpha)

    def _expectation_step_features(self, fsource, ftarget, y, sigma2, update_sigma2, objective_type, alpha):
        """filterreg.py:78-108 line by line on ``gaussian_filtering.Permutohedral`` (GPU lattice of any d <= 64)."""
        from . import gaussian_filtering as gf

        m, n = fsource.shape[0], ftarget.shape[0]
        sigma = np.sqrt(sigma2)
        fin = np.r_[fsource / sigma, ftarget / sigma]
        ph = gf.Permutohedral(fin)
        if ph.get_lattice_size() > n * alpha:  # :90-91
            ph = gf.Permutohedral(fin, False)
        zeros_m1 = np.zeros((m, 1))
        m0 = ph.filter(np.r_[zeros_m1, np.ones((n, 1))], m).flatten()[:m]
        m1 = ph.filter(np.r_[np.zeros((m, y.shape[1])), y], m)[:m]
        m2 = ph.filter(np.r_[zeros_m1, np.square(y).sum(axis=1)[:, None]], m).flatten()[:m] if update_sigma2 else None
        nx = None
        if objective_type == "pt2pl":
            if self._target_normals is None:
                raise ValueError("objective_type 'pt2pl' needs target_normals.")
            nx = ph.filter(np.r_[np.zeros((m, y.shape[1])), np.asarray(self._target_normals)], m)[:m]
        return EstepResult(m0, m1, m2, nx)

    def maximization_step(self, t_source, target, estep_res, w=0.0, objective_type="pt2pt"):
        """M-step on explicit E-step arrays (reference filterreg.py:110-113)."""
        return self._maximization_step(t_source, target, estep_res, self._tf_result, self._sigma2, w,
                                       objective_type=objective_type)

    @staticmethod
    @abc.abstractmethod
    def _maximization_step(t_source, target, estep_res, trans_p, sigma2, w=0.0, objective_type="pt2pt"):
        return None

    def registration(self, target, w=0.0, objective_type="pt2pt", maxiter=50, tol=0.001, min_sigma2=1.0e-4,
                     feature_fn=_identity):
        """EM driver (reference filterreg.py:120-147)."""
        assert self._tf_type is not None, "transformation type is None."
        if objective_type not in ("pt2pt", "pt2pl"):
            raise ValueError("Unknown objective_type: %s." % objective_type)
        if objective_type == "pt2pl" and self._target_normals is None:
            raise ValueError("objective_type 'pt2pl' needs target_normals.")
        target = _as_points(target)
        if self._source.shape[1] != target.shape[1] or target.shape[1] not in (2, 3):
            raise ValueError("source and target must both be (n, 2) or (n, 3) arrays.")
        if feature_fn is None:  # documented selector of the device-resident path
            feature_fn = _identity
        if feature_fn is not _identity:
            # the reference's own idiom `feature_fn=lambda x: x` (filterreg.py:121): a callable that hands the very
            # object it was given back, UNCHANGED, is the identity - no numerical probe on a sample (an extractor that needs
            # neighbourhoods would misbehave on one), just one call on a COPY of the whole source cloud (an in-place
            # extractor - `x -= x.mean(0); return x` - also returns its argument: it must neither be taken for the
            # identity nor touch the live source)
            probe = self._source.copy()
            if feature_fn(probe) is probe and np.array_equal(probe, self._source):
                feature_fn = _identity
        if feature_fn is not _identity:
            # any other callable takes the reference's loop (features on the host every iteration)
            return self._registration_features(target, w, objective_type, maxiter, tol, min_sigma2, feature_fn)
        q = None
        if self._sigma2 is None:
            self._sigma2 = max(mu.squared_kernel_sum(self._source, target), min_sigma2)
        plan = self._ensure_plan(target)
        plan.set_normals(self._target_normals if objective_type == "pt2pl" else None)
        dim = target.shape[1]
        res = MstepResult(self._tf_result, self._sigma2, None)
        # the transform and sigma2 live on the device: uploaded once, advanced by the M-step kernel
        plan.set_state(self._tf_result.rot, self._tf_result.t, self._sigma2)
        # tf_init_params['scale']: the reference's first transform applies it (filterreg.py:129) and every M-step
        # returns a scale-free RigidTransformation (:196), so it acts on the first iteration only
        scale0 = float(getattr(self._tf_result, "scale", 1.0))
        if scale0 != 1.0:
            plan.set_source(self._source * scale0)
        # Nobody looks at the intermediate results (no callbacks, no tolerance to test, no DEBUG log): the iterations are
        # only enqueued - one lattice-size hand-over per E-step, nothing read back - and the state is fetched once at the
        # end.  An iteration in which every m0 is zero leaves the device state alone (filterreg.py:136-138 would stop there;
        # the iterations after it repeat it and change nothing either).
        if not self._callbacks and tol < 0 and not log.isEnabledFor(10) and scale0 == 1.0 and maxiter > 0:
            for i in range(maxiter):
                plan.estep()
                plan.mstep(w, self._update_sigma2, objective_type, min_sigma2, read=False)
            out = plan.get_state()
            if out[19] > 0.0:
                rot = out[:9].reshape(3, 3)[:dim, :dim].copy()
                res = MstepResult(tf.RigidTransformation(rot, out[9:9 + dim].copy()), float(out[15]), float(out[18]))
                self._tf_result = res.transformation
                self._sigma2 = max(res.sigma2, min_sigma2)
                if out[16] == 0.0:  # the last iterations had nothing to fit: the driver's answer at that point
                    res = MstepResult(self._tf_result, self._sigma2, res.q)
            else:
                res = MstepResult(self._tf_result, self._sigma2, None)
            return res
        for i in range(maxiter):
            if i == 1 and scale0 != 1.0:
                plan.set_source(self._source)
            plan.estep()
            out = plan.mstep(w, self._update_sigma2, objective_type, min_sigma2)
            if out[16] == 0.0:  # every m0 == 0 (filterreg.py:167-168, :136-138)
                res = MstepResult(self._tf_result, self._sigma2, q)
                break
            rot = out[:9].reshape(3, 3)[:dim, :dim].copy()
            t = out[9:9 + dim].copy()
            res = MstepResult(tf.RigidTransformation(rot, t), float(out[15]), float(out[13]))
            self._tf_result = res.transformation
            self._sigma2 = max(res.sigma2, min_sigma2)
            for c in self._callbacks:
                c(self._tf_result)
            log.debug("Iteration: {}, Criteria: {}".format(i, res.q))
            if q is not None and abs(res.q - q) < tol:
                break
            q = res.q
        return res


    def _registration_features(self, target, w, objective_type, maxiter, tol, min_sigma2, feature_fn):
        """The reference's driver verbatim in structure (filterreg.py:120-147) for a non-identity ``feature_fn``:
        features are evaluated on the host every iteration, the lattice E-step and the M-step run on the GPU."""
        q = None
        ftarget = np.asarray(feature_fn(target), dtype=np.float64)
        if self._sigma2 is None:
            fsource = np.asarray(feature_fn(self._source), dtype=np.float64)
            self._sigma2 = max(mu.squared_kernel_sum(fsource, ftarget), min_sigma2)
        res = MstepResult(self._tf_result, self._sigma2, None)
        for i in range(maxiter):
            t_source = self._tf_result.transform(self._source)
            fsource = np.asarray(feature_fn(t_source), dtype=np.float64)
            estep_res = self.expectation_step(fsource, ftarget, target, self._sigma2, self._update_sigma2, objective_type)
            res = self.maximization_step(t_source, target, estep_res, w=w, objective_type=objective_type)
            if res.q is None:
                res = res._replace(q=q)
                break
            self._tf_result = res.transformation
            self._sigma2 = max(res.sigma2, min_sigma2)
            for c in self._callbacks:
                c(self._tf_result)
            log.debug("Iteration: {}, Criteria: {}".format(i, res.q))
            if q is not None and abs(res.q - q) < tol:
                break
            q = res.q
        return res


class RigidFilterReg(FilterReg):
    """Rigid FilterReg (reference filterreg.py:150-196)."""

    def __init__(self, source=None, target_normals=None, sigma2=None, update_sigma2=False, tf_init_params={}):
        super(RigidFilterReg, self).__init__(
            source=source, target_normals=target_normals, sigma2=sigma2, update_sigma2=update_sigma2
        )
        self._tf_type = tf.RigidTransformation
        self._tf_result = self._tf_type(**tf_init_params)
        if self._source is not None and "rot" not in tf_init_params and self._source.shape[1] == 2:
            # the reference needs explicit 2-D tf_init_params for 2-D data (examples/filterreg_rigid2d.py);
            # default to the 2-D identity instead of failing with a shape error
            self._tf_result = self._tf_type(np.identity(2), np.zeros(2))


    @staticmethod
    def _maximization_step(t_source, target, estep_res, trans_p, sigma2, w=0.0, objective_type="pt2pt"):
        """Rigid M-step from explicit arrays (reference filterreg.py:158-196): weighted Kabsch (pt2pt) or twist
        solve (pt2pl) on the GPU, composition with ``trans_p``, sigma2 re-estimated when ``estep_res.m2`` is given."""
        t_source = np.ascontiguousarray(_as_points(t_source))
        target = _as_points(target)
        m, dim = t_source.shape
        assert dim == 2 or dim == 3, "dim must be 2 or 3."
        if objective_type not in ("pt2pt", "pt2pl"):
            raise ValueError("Unknown objective_type: %s." % objective_type)
        m0, m1, m2, nx = estep_res
        if objective_type == "pt2pl" and nx is None:
            raise ValueError("objective_type 'pt2pl' needs estep_res.nx.")
        _lib.require_gpu()
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        m0, m1, m2 = f32(m0), f32(m1), f32(m2)
        nx = f32(nx) if objective_type == "pt2pl" else None
        rot = np.identity(3)
        rot[:dim, :dim] = trans_p.rot
        t = np.zeros(3)
        t[:dim] = trans_p.t
        dev, st = _current_device_and_stream()
        out = np.zeros(18)
        check(lib.prg_fr_mstep_from_arrays(dev, ctypes.c_void_p(st), ptr(t_source), m, dim, target.shape[0], ptr(m0),
                                           ptr(m1), ptr(m2), ptr(nx), ptr(np.ascontiguousarray(rot)), ptr(t),
                                           float(sigma2), float(w), ptr(out)))
        if out[16] == 0.0:  # every m0 == 0 (filterreg.py:167-168)
            return MstepResult(trans_p, sigma2, None)
        return MstepResult(tf.RigidTransformation(out[:9].reshape(3, 3)[:dim, :dim].copy(), out[9:9 + dim].copy()),
                           float(out[15]), float(out[13]))


def registration_filterreg(source, target, target_normals=None, sigma2=None, update_sigma2=False, w=0,
                           objective_type="pt2pt", maxiter=50, tol=0.001, min_sigma2=1.0e-4, feature_fn=_identity,
                           callbacks=[], **kwargs):
    """One-call rigid FilterReg with the reference's signature (filterreg.py:269-317).

    source, target  : (n, 2|3) arrays or Open3D point clouds
    target_normals  : (N, 3) normals, required for ``objective_type='pt2pl'``
    sigma2          : kernel variance (None = automatic); update_sigma2 re-estimates it every iteration
    w               : outlier mass in [0, 1); objective_type 'pt2pt' (Kabsch) or 'pt2pl' (twist)
    maxiter, tol    : stop after maxiter iterations or when |q - q_prev| < tol; min_sigma2 clamps the variance
    feature_fn      : maps (n, D) points to (n, d <= 64) features the lattice is built over (identity by default; a
                      host callable, evaluated once per iteration); callbacks get the transformation after every iteration
    **kwargs        : ``tf_init_params`` for the starting rigid transform
    Returns ``MstepResult(transformation, sigma2, q)``.
    """
    frg = RigidFilterReg(_as_points(source), _as_points(target_normals), sigma2, update_sigma2, **kwargs)
    frg.set_callbacks(callbacks)
    return frg.registration(
        _as_points(target), w=w, objective_type=objective_type, maxiter=maxiter, tol=tol, min_sigma2=min_sigma2,
        feature_fn=feature_fn,
    )


def kabsch(model, target, weight):
    """Weighted Kabsch on the GPU (reference ``_kabsch.kabsch`` / ``kabsch2d``, cc/kabsch.cc:6-109)."""
    _lib.require_gpu()
    model = np.ascontiguousarray(model, dtype=np.float32)
    target = np.ascontiguousarray(target, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    dim = model.shape[1]
    dev, st = _current_device_and_stream()
    rot = np.empty((dim, dim))
    t = np.empty(dim)
    check(lib.prg_kabsch_weighted(dev, ctypes.c_void_p(st), ptr(model), ptr(target), ptr(weight), model.shape[0], dim,
                                  ptr(rot), ptr(t)))
    return rot, t
