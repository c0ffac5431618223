"""Coherent Point Drift on MI355X - drop-in for ``probreg.cpd`` (reference probreg/cpd.py).

Same public surface as the reference: ``EstepResult`` / ``MstepResult`` (cpd.py:17-18),
``CoherentPointDrift`` with ``set_source`` / ``set_callbacks`` / ``expectation_step`` /
``maximization_step`` / ``registration`` (cpd.py:29-120), ``RigidCPD`` (:123-192), ``AffineCPD``
(:195-244), ``NonRigidCPD`` (:247-303) and ``registration_cpd`` (:407-456).

What is different underneath: the E-step (M x N responsibilities, P1, Pt1, PX), the sigma2
initialiser, the transform of the source and the M-step all run in ``libprobreg_hip.so`` on the
GPU; the M x N matrix is never materialised.  ``use_cuda`` is accepted and ignored (the engine is
always the HIP one; there is no NumPy path).  Under ``torchrun`` (torch.distributed initialised)
the target cloud is sharded over the ranks and one all-reduce per iteration combines the moments.
"""
import abc
from collections import namedtuple

import numpy as np

from . import _lib
from . import dist as pdist
from . import transformation as tf
from .engine import CpdPlan
from .log import log

EstepResult = namedtuple("EstepResult", ["pt1", "p1", "px", "n_p"])
MstepResult = namedtuple("MstepResult", ["transformation", "sigma2", "q"])
MstepResult.__doc__ = """What an M-step (and ``registration``) hands back - same 3-tuple as the reference (cpd.py:18).

    transformation : the estimated source -> target map (a ``probreg_amd.transformation`` object)
    sigma2         : current isotropic variance of the GMM components
    q              : value of the objective the convergence test looks at
"""


def _as_points(x):
    """ndarray or Open3D PointCloud (duck-typed ``.points``) -> float64 ndarray (cpd.py:444)."""
    if hasattr(x, "points") and not isinstance(x, np.ndarray):
        x = x.points
    return np.asarray(x, dtype=np.float64)


def _params_block(linear, t, scale, dim, delta=None):
    p = np.zeros(16)
    lin = np.identity(3)
    lin[:dim, :dim] = np.asarray(linear, dtype=np.float64)
    p[:9] = lin.ravel()
    p[9:9 + dim] = np.asarray(t, dtype=np.float64)
    p[12] = scale
    if delta is not None:
        p[13:13 + dim] = np.asarray(delta, dtype=np.float64)
    return p


class CoherentPointDrift(abc.ABC):
    """Base class of the three CPD flavours (reference cpd.py:29-120): owns the GPU plan and the EM driver,
    subclasses supply the initialisation and the M-step.

    source   : (M, D) array, the cloud that gets moved (may also be given later through ``set_source``)
    use_cuda : kept so existing call sites run unchanged; the engine is always the HIP one
    device   : GPU index; defaults to the calling process's current torch device
    """

    _kind = None

    def __init__(self, source=None, use_cuda=False, device=None):
        self._source = None if source is None else _as_points(source)
        self._tf_type = None
        self._callbacks = []
        self._device = device
        self._plan = None
        self.xp = np

    # -- reference API ------------------------------------------------------------------------
    def set_source(self, source):
        self._source = _as_points(source)

    def set_callbacks(self, callbacks):
        self._callbacks.extend(callbacks)

    def expectation_step(self, t_source, target, sigma2, w=0.0):
        """Expectation step for CPD (reference cpd.py:71-88) on explicit arrays.

        Runs the two GPU pair sweeps on ``t_source`` as given (identity transform) and returns the
        reference's ``EstepResult(pt1, p1, px, n_p)`` as float64 arrays.
        """
        t_source = np.asarray(t_source)
        target = np.asarray(target)
        assert t_source.ndim == 2 and target.ndim == 2, "source and target must have 2 dimensions."
        # centre in fp64 before the fp32 upload (pairwise differences are translation invariant)
        c = target.mean(axis=0)
        plan = CpdPlan(self._device)
        try:
            plan.set_source(t_source - c)
            plan.set_target(target - c)
            p = np.zeros(_lib.PRG_NPARAMS)
            p[:13] = _params_block(np.identity(t_source.shape[1]), np.zeros(t_source.shape[1]), 1.0,
                                   t_source.shape[1])[:13]
            p[13] = sigma2
            plan.set_params(p)
            plan.estep(w)
            pt1, p1, px = plan.get_estep()
        finally:
            plan.close()
        px = px + np.outer(p1, c)  # undo the centring: sum_n P x_n = sum_n P (x_n - c) + p1 c
        return EstepResult(pt1, p1, px, float(np.sum(p1)))

    def maximization_step(self, target, estep_res, sigma2_p=None):
        return self._maximization_step(self._source, target, estep_res, sigma2_p)

    @staticmethod
    @abc.abstractmethod
    def _maximization_step(source, target, estep_res, sigma2_p=None, xp=np):
        """A static method in every subclass, with the reference's signature (cpd.py:95-104)."""
        return None

    # -- plan handling ------------------------------------------------------------------------
    def _centres(self, source, target_full):
        """fp64 centres subtracted before the fp32 upload (input conditioning)."""
        return source.mean(axis=0), target_full.mean(axis=0)

    def _setup_plan(self, target):
        """Upload clouds (target sharded over ranks), run the sigma2 initialiser. Returns the plan."""
        source = self._source
        target = _as_points(target)
        assert source.ndim == 2 and target.ndim == 2, "source and target must have 2 dimensions."
        if source.shape[1] != target.shape[1] or source.shape[1] not in (2, 3):
            raise ValueError("source and target must both be (n, 2) or (n, 3) arrays.")
        rank, world = pdist.world()
        # the GLOBAL condition, so every rank raises together (a per-rank "my slice is empty" test would leave the
        # other ranks waiting in the first all-reduce)
        if target.shape[0] < world:
            raise ValueError("the target has %d points, fewer than the %d ranks it is sharded over."
                             % (target.shape[0], world))
        cy, cx = self._centres(source, target)
        rows = pdist.spatial_shard(target, rank, world)
        plan = self._plan if self._plan is not None else CpdPlan(self._device)
        self._plan = plan
        self._cy, self._cx = cy, cx
        if not getattr(self, "_source_uploaded", False):
            plan.set_source(source - cy)
        plan.set_target(target[rows] - cx, n_global=target.shape[0])
        # the per-iteration collective: issued by the library itself on the plan's stream (RCCL, prg_cpd_set_comm) when the
        # process group is an nccl one; through torch.distributed on the bound moment tensor otherwise (gloo)
        comm = pdist.native_comm(plan.device) if hasattr(plan, "set_comm") else None
        if hasattr(plan, "set_comm") and getattr(plan, "_comm", None) is not comm:
            plan.set_comm(comm)  # (also None: a reused plan must not keep a communicator that is gone or no longer wanted)
        mom = plan.moments_tensor() if comm is None and pdist.initialized() else None
        plan.init_sums()
        if mom is not None:
            pdist.all_reduce_sum_(mom, getattr(plan, "stream", None))
        return plan

    def _all_reduce_moments(self, plan):
        if getattr(plan, "_comm", None) is not None:
            return  # done inside prg_cpd_estep
        if plan._moments_tensor is not None:
            pdist.all_reduce_sum_(plan._moments_tensor, getattr(plan, "stream", None))

    def _restart(self):
        """Put an initialised plan back to the start of its registration (sigma2 initialiser, q0, initial transform)
        without re-uploading the clouds - bench.py times the SAME EM iterations whatever its warm-up did."""
        plan = self._plan
        mom = plan.moments_tensor() if getattr(plan, "_comm", None) is None and pdist.initialized() else None
        plan.init_sums()
        if mom is not None:
            pdist.all_reduce_sum_(mom, getattr(plan, "stream", None))
        plan.init_params(self._init_block)
        return plan

    @abc.abstractmethod
    def _initialize(self, target):
        return MstepResult(None, None, None)

    @abc.abstractmethod
    def _device_mstep(self, plan):
        pass

    def _iterate_native(self, plan, w, n_iter):
        """Rigid / affine: the whole loop enqueued inside the library (prg_cpd_iterate) when nothing between the E-step and
        the M-step has to go through Python - one rank, or the library's own RCCL all-reduce.  False: not taken."""
        kind = getattr(self, "_kind", None)
        if kind not in (_lib.PRG_TF_RIGID, _lib.PRG_TF_AFFINE) or not hasattr(plan, "iterate"):
            return False
        if getattr(plan, "_comm", None) is None and pdist.initialized():
            return False  # the collective is torch.distributed's: one Python round trip per iteration
        plan.iterate(kind, getattr(self, "_update_scale", True), w, n_iter)
        return True

    @abc.abstractmethod
    def _result_from_params(self, params):
        pass

    def registration(self, target, w=0.0, maxiter=50, tol=0.001):
        """EM driver (reference cpd.py:106-120); the loop body runs on the GPU.

        The host reads the 32-double parameter block back only when it has to: every iteration if
        there are callbacks, DEBUG logging or a non-negative ``tol`` (the convergence test needs
        ``q``), otherwise once at the end.
        """
        assert self._tf_type is not None, "transformation type is None."
        res = self._initialize(target)
        plan = self._plan
        q = res.q
        need_host = bool(self._callbacks) or tol >= 0 or log.isEnabledFor(10)
        if not need_host and maxiter > 0 and self._iterate_native(plan, w, maxiter):
            return self._result_from_params(plan.get_params())
        # a rigid iteration consumes nothing of its E-step but the 23 moments (callbacks see the transformation only): the
        # dense regime may run as ONE sweep over the pairs (prg_cpd_set_moments_only; the per-point p1 / px of the public
        # expectation_step come from a plan of their own)
        fuse = getattr(self, "_kind", None) == _lib.PRG_TF_RIGID and hasattr(plan, "set_moments_only")
        if fuse:
            plan.set_moments_only(1)
        try:
            return self._em_loop(plan, w, maxiter, tol, need_host, res, q)
        finally:
            if fuse:
                plan.set_moments_only(0)

    def _em_loop(self, plan, w, maxiter, tol, need_host, res, q):
        for i in range(maxiter):
            plan.estep(w)
            self._all_reduce_moments(plan)
            self._device_mstep(plan)
            if need_host:
                res = self._result_from_params(plan.get_params())
                for c in self._callbacks:
                    c(res.transformation)
                log.debug("Iteration: {}, Criteria: {}".format(i, res.q))
                if abs(res.q - q) < tol:
                    break
                q = res.q
        if not need_host and maxiter > 0:
            res = self._result_from_params(plan.get_params())
        return res


class RigidCPD(CoherentPointDrift):
    """Rigid (rotation, translation, optional isotropic scale) CPD - reference cpd.py:123-192.

    update_scale   : estimate the scale as well (True, the reference's default) or pin it to 1
    tf_init_params : optional starting transform, keys ``rot`` / ``t`` / ``scale`` as in
                     ``RigidTransformation``
    """

    _kind = _lib.PRG_TF_RIGID

    def __init__(self, source=None, update_scale=True, tf_init_params={}, use_cuda=False, device=None):
        super(RigidCPD, self).__init__(source, use_cuda, device)
        self._tf_type = tf.RigidTransformation
        self._update_scale = update_scale
        self._tf_init_params = tf_init_params

    def _initialize(self, target):
        dim = self._source.shape[1]
        plan = self._setup_plan(target)
        ip = dict(self._tf_init_params)
        ip.pop("xp", None)
        rot = np.asarray(ip.get("rot", np.identity(dim)), dtype=np.float64)
        t = np.asarray(ip.get("t", np.zeros(dim)), dtype=np.float64)
        scale = float(ip.get("scale", 1.0))
        # centred frame: z - cx = s R (y - cy) + t'  with  t' = t + s R cy - cx
        t_c = t + scale * rot @ self._cy - self._cx
        self._init_block = _params_block(rot, t_c, scale, dim, self._cx - self._cy)
        plan.init_params(self._init_block)
        return self._result_from_params(plan.get_params())

    def _device_mstep(self, plan):
        plan.mstep(_lib.PRG_TF_RIGID, self._update_scale)

    def _result_from_params(self, params):
        return _rigid_result(params, self._source.shape[1], self._cy, self._cx)

    def maximization_step(self, target, estep_res, sigma2_p=None):
        return self._maximization_step(self._source, target, estep_res, sigma2_p, self._update_scale,
                                       _device=self._device)

    @staticmethod
    def _maximization_step(source, target, estep_res, sigma2_p=None, update_scale=True, xp=np, _device=None):
        """Rigid M-step from explicit EstepResult arrays, on the GPU - a static method with the reference's signature
        (cpd.py:160-192); ``xp`` is accepted and ignored, ``_device`` picks the GPU (default: the current one)."""
        return _mstep_from_arrays(source, target, estep_res, _lib.PRG_TF_RIGID, update_scale, _device)


class AffineCPD(CoherentPointDrift):
    """Affine CPD: full D x D matrix ``b`` plus translation - reference cpd.py:195-244.  ``tf_init_params`` may hold
    ``b`` / ``t`` to start from."""

    _kind = _lib.PRG_TF_AFFINE

    def __init__(self, source=None, tf_init_params={}, use_cuda=False, device=None):
        super(AffineCPD, self).__init__(source, use_cuda, device)
        self._tf_type = tf.AffineTransformation
        self._tf_init_params = tf_init_params

    def _initialize(self, target):
        dim = self._source.shape[1]
        plan = self._setup_plan(target)
        ip = dict(self._tf_init_params)
        ip.pop("xp", None)
        b = np.asarray(ip.get("b", np.identity(dim)), dtype=np.float64)
        t = np.asarray(ip.get("t", np.zeros(dim)), dtype=np.float64)
        t_c = t + b @ self._cy - self._cx
        self._init_block = _params_block(b, t_c, 1.0, dim, self._cx - self._cy)
        plan.init_params(self._init_block)
        return self._result_from_params(plan.get_params())

    def _device_mstep(self, plan):
        plan.mstep(_lib.PRG_TF_AFFINE, True)

    def _result_from_params(self, params):
        return _affine_result(params, self._source.shape[1], self._cy, self._cx)

    def maximization_step(self, target, estep_res, sigma2_p=None):
        return self._maximization_step(self._source, target, estep_res, sigma2_p, _device=self._device)

    @staticmethod
    def _maximization_step(source, target, estep_res, sigma2_p=None, xp=np, _device=None):
        """Affine M-step from explicit EstepResult arrays, on the GPU - static, the reference's signature
        (cpd.py:219-244)."""
        return _mstep_from_arrays(source, target, estep_res, _lib.PRG_TF_AFFINE, True, _device)


def _rigid_result(params, dim, cy, cx):
    """Parameter block of the centred frame -> MstepResult in the caller's frame (z - cx = s R (y - cy) + t')."""
    rot = params[:9].reshape(3, 3)[:dim, :dim].copy()
    scale = float(params[12])
    t = params[9:9 + dim] + cx - scale * rot @ cy
    return MstepResult(tf.RigidTransformation(rot, t, scale), float(params[13]), float(params[14]))


def _affine_result(params, dim, cy, cx):
    b = params[:9].reshape(3, 3)[:dim, :dim].copy()
    if not np.all(np.isfinite(b)):
        # Y^T diag(P1) Y is singular (fewer than D + 1 supported source points, or coplanar ones): the
        # reference's np.linalg.solve (cpd.py:237) raises here, the device solve leaves non-finite numbers
        raise np.linalg.LinAlgError("Singular matrix")
    t = params[9:9 + dim] + cx - b @ cy
    return MstepResult(tf.AffineTransformation(b, t), float(params[13]), float(params[14]))


def _mstep_from_arrays(source, target, estep_res, kind, update_scale, device=None):
    source = _as_points(source)
    target = _as_points(target)
    pt1, p1, px, n_p = estep_res
    cy, cx = source.mean(axis=0), target.mean(axis=0)
    plan = CpdPlan(device)
    try:
        plan.set_source(source - cy)
        plan.set_target(target - cx)
        # moments in the centred frame: px' = px - p1 cx
        plan.moments_from_estep(pt1, p1, np.asarray(px) - np.outer(p1, cx))
        p = np.zeros(_lib.PRG_NPARAMS)
        p[0] = p[4] = p[8] = p[12] = 1.0
        p[13] = 1.0
        plan.set_params(p)
        plan.mstep(kind, update_scale)
        params = plan.get_params()
    finally:
        plan.close()
    conv = _rigid_result if kind == _lib.PRG_TF_RIGID else _affine_result
    return conv(params, source.shape[1], cy, cx)


class NonRigidCPD(CoherentPointDrift):
    """Non-rigid CPD (motion coherence): displacement field ``G W`` over the source points - reference
    cpd.py:247-303.

    beta : width parameter of the Gaussian kernel ``G = exp(-d^2 / (2 beta))`` (note: beta, not beta^2)
    lmd  : weight of the smoothness regulariser
    The kernel is factorised on the GPU when the source is set (``G = F F^T`` by pivoted Cholesky, exact to 1e-11 per entry; the
    M x M matrix itself is only kept when its rank is too high for that - see ``prg_cpd_nonrigid_build_g``).
    """

    _kind = _lib.PRG_TF_NONRIGID
    _solver_mode = 1  # prg_cpd_nonrigid_set_solver: 1 = low-rank factor when the rank allows, 0 = always the dense matrix
    _factor_tol = 0.0  # largest entry of G - F F^T the factor may leave (0: the library's default, 1e-11)

    def __init__(self, source=None, beta=2.0, lmd=2.0, use_cuda=False, device=None):
        super(NonRigidCPD, self).__init__(source, use_cuda, device)
        self._tf_type = tf.NonRigidTransformation
        self._beta = beta
        self._lmd = lmd
        self._tf_obj = None
        if self._source is not None:
            self._build()

    def _centres(self, source, target_full):
        # one common origin for both clouds (G W is a displacement field over the source's own frame)
        return self._origin, self._origin

    def _build(self):
        self._plan, self._origin = _nonrigid_plan(self._source, self._beta, self._device, self._solver_mode, self._factor_tol)
        self._source_uploaded = True
        self._tf_obj = tf.NonRigidTransformation(None, self._source, self._beta, _plan=self._plan,
                                                 _plan_points=self._source - self._origin)

    def set_source(self, source):
        self._source = _as_points(source)
        self._build()

    def _initialize(self, target):
        plan = self._setup_plan(target)
        self._init_block = None
        plan.init_params(None)
        self._tf_obj.w = np.zeros_like(self._source)
        plan.set_w(self._tf_obj.w)
        return self._result_from_params(plan.get_params())

    def _all_reduce_moments(self, plan):
        super(NonRigidCPD, self)._all_reduce_moments(plan)
        if pdist.initialized() and getattr(plan, "_comm", None) is None:
            self._all_reduce_rowacc(plan)

    def _all_reduce_rowacc(self, plan):
        import torch

        ptr_, count = plan.rowacc_tensor_view()

        class _View(object):
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr_, False), "version": 2}

        t = torch.as_tensor(_View(), device="cuda:%d" % plan.device)
        pdist.all_reduce_sum_(t, plan.stream)

    def _device_mstep(self, plan):
        plan.mstep_nonrigid(self._lmd)

    def _result_from_params(self, params):
        if params[16] > 0:
            self._tf_obj.w = self._plan.get_w()
        return MstepResult(self._tf_obj, float(params[13]), float(params[14]))

    def maximization_step(self, target, estep_res, sigma2_p=None):
        return self._maximization_step(self._source, target, estep_res, sigma2_p, self._tf_obj, self._lmd,
                                       _device=self._device, _solver_mode=self._solver_mode)

    @staticmethod
    def _maximization_step(source, target, estep_res, sigma2_p, tf_obj, lmd, xp=np, _priors=None, _device=None,
                           _solver_mode=1):
        """Non-rigid M-step from explicit EstepResult arrays, on the GPU - a static method with the reference's
        signature (cpd.py:284-303): any ``source`` with a ``NonRigidTransformation`` built on it will do.

        ``(diag(p1) G + lmd sigma2_p I) W = px - diag(p1) Y`` is solved on a plan that belongs to ``tf_obj`` (created
        on the first call, kept for the next ones: the kernel factor of ``tf_obj``'s control points) - never on the
        plan of a running registration, whose EM state a call from a callback must not disturb.  Like the reference,
        ``tf_obj.w`` is updated in place, the object is returned and ``q`` is the new sigma2.  ``sigma2_p`` is the
        variance of the E-step that produced ``estep_res``.
        """
        if sigma2_p is None:
            raise ValueError("NonRigidCPD.maximization_step needs sigma2_p (the regulariser is lmd * sigma2_p).")
        source = _as_points(source)
        target = _as_points(target)
        ctrl = np.asarray(tf_obj._points, dtype=np.float64)
        if source.shape != ctrl.shape or not np.array_equal(source, ctrl):
            raise ValueError("NonRigidCPD._maximization_step: source must be the control points tf_obj was built on "
                             "(the reference reads G from tf_obj and Y from source; both are the same cloud).")
        key = (_device, _solver_mode, float(tf_obj._beta))
        cache = getattr(tf_obj, "_mstep_plan", None)
        if cache is not None and cache[2] != key:  # another device / solver / kernel width: the old factor is void
            cache[0].close()
            cache = None
        if cache is None:
            cache = _nonrigid_plan(ctrl, tf_obj._beta, _device, _solver_mode) + (key,)
            tf_obj._mstep_plan = cache  # released by NonRigidTransformation.close() / __del__
        plan, origin = cache[0], cache[1]
        pt1, p1, px, _n_p = estep_res
        plan.set_target(target - origin, n_global=target.shape[0])
        if _priors is not None:
            alpha, p1_tilde, px_tilde = _priors
            plan.set_priors(p1_tilde, np.asarray(px_tilde, dtype=np.float64) - np.outer(p1_tilde, origin), alpha)
        plan.moments_from_estep(pt1, p1, np.asarray(px, dtype=np.float64) - np.outer(p1, origin))  # P X in the plan's frame
        p = plan.get_params()
        p[13] = float(sigma2_p)
        plan.set_params(p)
        plan.mstep_nonrigid(lmd)
        out = plan.get_params()
        tf_obj.w = plan.get_w()
        return MstepResult(tf_obj, float(out[13]), float(out[14]))


def _nonrigid_plan(points, beta, device=None, solver_mode=1, factor_tol=0.0):
    """A GPU plan holding the kernel ``G = exp(-|y_i - y_j|^2 / (2 beta))`` of ``points`` (as its factor when the rank
    allows).  Returns ``(plan, origin)``; the plan works in the frame ``points - origin``."""
    plan = CpdPlan(device)
    # both clouds are Morton-sorted inside the plan (culled / matrix-core sweeps); W, the priors and the transformed
    # points cross the C-ABI in the caller's order, and every rank sorts the replicated source the same way, so the
    # per-point all-reduce block lines up across ranks
    plan.set_options(sort_source=True, sort_target=True, cull=True)
    # G is built from the float32 source exactly as the reference's pybind cast sees it (cc/math_utils.cc:17-19),
    # so clouds near the origin are uploaded as they are.  A cloud FAR from the origin (coordinates many times its
    # own extent, e.g. examples/face-x.txt with z ~ 1277) would lose its fine structure in that cast - in the
    # E-step's float32 differences as well as in G - so it is shifted by its fp64 mean first: differences, G and the
    # displacement field are translation invariant, and the reference's own float32 G is no yardstick there.
    c = points.mean(axis=0)
    ext = float(np.max(points.max(axis=0) - points.min(axis=0)))
    origin = c if float(np.max(np.abs(c))) > 8.0 * max(ext, 1e-300) else np.zeros(points.shape[1])
    plan.set_source(points - origin)
    plan.set_nonrigid_solver(solver_mode, 0, factor_tol)
    plan.build_g(beta)
    return plan, origin


class ConstrainedNonRigidCPD(NonRigidCPD):
    """Non-rigid CPD with known correspondences (reference cpd.py:306-404; Golyanik et al., "Extended coherent
    point drift algorithm with correspondence priors and optimal subsampling", 2016).

    On top of ``NonRigidCPD``'s ``source`` / ``beta`` / ``lmd``:

    alpha                  : how much the listed correspondences are trusted - the prior terms enter the linear
                             system scaled by ``sigma2 / alpha``, so 1e-8 pins the pairs and 1 barely uses them
    idx_source, idx_target : equally long integer arrays; source point ``idx_source[k]`` is known to match target
                             point ``idx_target[k]``
    use_cuda               : kept for call-site compatibility, ignored (the engine is always the HIP one)

    The reference materialises a dense M x N 0-1 matrix ``p_tilde`` (cpd.py:370-374); only its row sums
    ``p1_tilde`` and ``px_tilde = p_tilde @ target`` enter the M-step, so they are formed directly from the
    index lists here.
    """

    def __init__(self, source=None, beta=2.0, lmd=2.0, alpha=1e-8, use_cuda=False, idx_source=None, idx_target=None,
                 device=None):
        self.alpha = alpha
        self.idx_source, self.idx_target = idx_source, idx_target
        super(ConstrainedNonRigidCPD, self).__init__(source, beta, lmd, use_cuda, device)

    def _priors_for(self, target):
        """``p1_tilde`` (M) and ``px_tilde`` (M x D) of the reference's 0-1 matrix ``p_tilde`` (cpd.py:370-374)."""
        target = _as_points(target)
        m, dim = self._source.shape
        self.p1_tilde = np.zeros(m)
        self.px_tilde = np.zeros((m, dim))
        if self.idx_source is not None and self.idx_target is not None:
            # p_tilde[idx_source, idx_target] = 1 is an assignment: duplicate pairs count once (cpd.py:372-373)
            pairs = np.unique(np.stack([np.asarray(self.idx_source), np.asarray(self.idx_target)], axis=1), axis=0)
            np.add.at(self.p1_tilde, pairs[:, 0], 1.0)
            np.add.at(self.px_tilde, pairs[:, 0], target[pairs[:, 1]])
        return self.p1_tilde, self.px_tilde

    def _initialize(self, target):
        res = super(ConstrainedNonRigidCPD, self)._initialize(target)
        p1_tilde, px_tilde = self._priors_for(target)
        # (the plan works in the shifted frame of _nonrigid_plan)
        self._plan.set_priors(p1_tilde, px_tilde - np.outer(p1_tilde, self._origin), self.alpha)
        return res

    def maximization_step(self, target, estep_res, sigma2_p=None):
        # the reference reads self.p1_tilde / self.px_tilde left by _initialize (cpd.py:349-364); here they are (re)built
        # from ``target``, so the call also works on an object that has not run a registration yet
        p1_tilde, px_tilde = self._priors_for(target)
        return self._maximization_step(self._source, target, estep_res, sigma2_p, self._tf_obj, self._lmd, self.alpha,
                                       p1_tilde, px_tilde, _device=self._device, _solver_mode=self._solver_mode)

    @staticmethod
    def _maximization_step(source, target, estep_res, sigma2_p, tf_obj, lmd, alpha, p1_tilde, px_tilde, xp=np,
                           _device=None, _solver_mode=1):
        """Constrained non-rigid M-step on explicit arrays - static, the reference's signature (cpd.py:377-404)."""
        return NonRigidCPD._maximization_step(source, target, estep_res, sigma2_p, tf_obj, lmd, xp,
                                              _priors=(alpha, p1_tilde, px_tilde), _device=_device,
                                              _solver_mode=_solver_mode)


def registration_cpd(source, target, tf_type_name="rigid", w=0.0, maxiter=50, tol=0.001, callbacks=[],
                     use_cuda=False, **kwargs):
    """One-call CPD registration with the reference's signature (cpd.py:407-456).

    source, target : (n, 2|3) arrays or Open3D point clouds (anything with a ``.points`` attribute)
    tf_type_name   : 'rigid' | 'affine' | 'nonrigid' | 'nonrigid_constrained'
    w              : mass of the uniform outlier component, in [0, 1)
    maxiter, tol   : EM stops after ``maxiter`` iterations or when |q - q_prev| < tol (use tol < 0 for a fixed count)
    callbacks      : callables invoked with the current transformation after every iteration
    use_cuda       : ignored (kept for drop-in compatibility)
    **kwargs       : forwarded to the class: ``update_scale`` / ``tf_init_params`` (rigid, affine), ``beta`` / ``lmd``
                     (non-rigid), ``alpha`` / ``idx_source`` / ``idx_target`` (constrained)
    Returns ``MstepResult(transformation, sigma2, q)``.  Under torchrun the target is sharded over the ranks.
    """
    if tf_type_name == "rigid":
        cpd = RigidCPD(_as_points(source), use_cuda=use_cuda, **kwargs)
    elif tf_type_name == "affine":
        cpd = AffineCPD(_as_points(source), use_cuda=use_cuda, **kwargs)
    elif tf_type_name == "nonrigid":
        cpd = NonRigidCPD(_as_points(source), use_cuda=use_cuda, **kwargs)
    elif tf_type_name == "nonrigid_constrained":
        cpd = ConstrainedNonRigidCPD(_as_points(source), use_cuda=use_cuda, **kwargs)
    else:
        raise ValueError("Unknown transformation type %s" % tf_type_name)
    cpd.set_callbacks(callbacks)
    return cpd.registration(_as_points(target), w, maxiter, tol)
