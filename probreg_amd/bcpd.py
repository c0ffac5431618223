"""Bayesian Coherent Point Drift on MI355X - drop-in for ``probreg.bcpd`` (reference probreg/bcpd.py).

Public surface as in the reference: ``EstepResult`` / ``MstepResult`` (bcpd.py:17-18),
``BayesianCoherentPointDrift`` (:31-98), ``CombinedBCPD`` (:101-151), ``registration_bcpd`` (:154-185).

What runs where
  * E-step (bcpd.py:53-72): the same two fused pair sweeps as CPD in ``libprobreg_hip.so``; the per-source factor
    ``alpha_m exp(-s^2 D Sigma_mm / 2 sigma2)`` enters as an additive squared distance, so the M x N matrix, its
    ``np.kron`` blow-ups and the Python list comprehension of the reference never exist.
  * M-step (bcpd.py:123-151): ``Sigma = (lmd G^-1 + c diag(nu))^-1`` is needed only through ``diag(Sigma)`` and
    ``Sigma diag(nu) R``; both come from one fp64 Cholesky + triangular solve on the GPU (Woodbury form, no
    ``G^-1``, no M x M inverse, no (MD) x (MD) Kronecker matrices) - ``prg_cpd_bcpd_solve``.  The remaining O(M + N)
    algebra (digamma, D x D SVD, sigma2) is numpy on the host, as in the reference.
  * Convergence criterion (bcpd.py:93): brute-force nearest neighbours on the GPU instead of a cKDTree.

Differences a caller can see: ``MstepResult.sigma_mat`` is ``diag(Sigma)`` (length M) - ``expectation_step`` accepts
either that or a full matrix, of which it reads the diagonal exactly like the reference; where the reference would
divide by ``nu_m == 0`` (NaN) the point simply gets no pull.  With ``torch.distributed`` initialised every rank
solves the whole problem (replicas; the M-step is not sharded).
"""
import abc
from collections import namedtuple

import numpy as np
import scipy.special as spsp

from . import _lib
from . import math_utils as mu
from . import transformation as tf
from .cpd import _as_points, _params_block
from .engine import CpdPlan
from .log import log

EstepResult = namedtuple("EstepResult", ["nu_d", "nu", "n_p", "px", "x_hat"])
MstepResult = namedtuple("MstepResult", ["transformation", "u_hat", "sigma_mat", "alpha", "sigma2"])
MstepResult.__doc__ = """State after a BCPD M-step (reference bcpd.py:18): the CombinedTransformation, u_hat = y + v_hat,
    diag(Sigma), the mixing weights alpha (next E-step inputs) and the new sigma2."""

_TINY = np.finfo(np.float64).tiny


def _sigma_diag(sigma_mat, m):
    s = np.asarray(sigma_mat, dtype=np.float64)
    if s.ndim == 2:
        s = np.diag(s)
    if s.ndim == 0:
        s = np.full(m, float(s))
    assert s.shape == (m,), "sigma_mat must be M x M or its length-M diagonal"
    return s


def _estep_on_plan(plan, n_target, dim, scale, alpha, sigma_diag, sigma2, w):
    """Weights -> GPU E-step -> (nu_d, nu, px) in the plan's (centred) frame.  bcpd.py:56-67."""
    lw = np.log(np.maximum(np.broadcast_to(np.asarray(alpha, dtype=np.float64), (plan.m,)), _TINY))
    lw = lw - (scale ** 2) * dim / (2.0 * sigma2) * sigma_diag
    top = float(lw.max())
    # P is invariant to a common factor of the weights as long as the uniform term is divided by it too
    ratio = min(np.exp(min(-top, 700.0)) / n_target, 1e300)
    plan.set_source_weights(lw - top, ratio)
    plan.estep(w)
    return plan.get_estep()


class BayesianCoherentPointDrift(abc.ABC):
    """EM driver of BCPD (reference bcpd.py:31-98); ``device`` picks the GPU (default: current torch device)."""

    def __init__(self, source=None, device=None):
        self._source = None if source is None else _as_points(source)
        self._tf_type = None
        self._callbacks = []
        self._device = device
        self._plan = None

    def set_source(self, source):
        self._source = _as_points(source)
        self._close_plan()

    def set_callbacks(self, callbacks):
        self._callbacks.extend(callbacks)

    def _close_plan(self):
        if self._plan is not None:
            self._plan.close()
            self._plan = None

    def __del__(self):
        try:
            self._close_plan()
        except Exception:  # pragma: no cover - interpreter shutdown
            pass

    @abc.abstractmethod
    def _initialize(self, target):
        return MstepResult(None, None, None, None, None)

    def expectation_step(self, t_source, target, scale, alpha, sigma_mat, sigma2, w=0.0):
        """BCPD E-step on explicit arrays (reference bcpd.py:53-72): EstepResult(nu_d, nu, n_p, px, x_hat)."""
        t_source = np.asarray(t_source, dtype=np.float64)
        target = np.asarray(target, dtype=np.float64)
        assert t_source.ndim == 2 and target.ndim == 2, "source and target must have 2 dimensions."
        m, dim = t_source.shape
        c = target.mean(axis=0)
        plan = CpdPlan(self._device)
        try:
            plan.set_source(t_source - c)
            plan.set_target(target - c)
            p = np.zeros(_lib.PRG_NPARAMS)
            p[:13] = _params_block(np.identity(dim), np.zeros(dim), 1.0, dim)[:13]
            p[13] = sigma2
            plan.set_params(p)
            nu_d, nu, px = _estep_on_plan(plan, target.shape[0], dim, scale, alpha, _sigma_diag(sigma_mat, m), sigma2, w)
        finally:
            plan.close()
        px = px + np.outer(nu, c)
        return EstepResult(nu_d, nu, float(np.sum(nu)), px, px / np.maximum(nu, _TINY)[:, None])

    def maximization_step(self, target, estep_res, sigma2_p=None):
        return self._maximization_step(self._source, target, estep_res, sigma2_p)

    @staticmethod
    @abc.abstractmethod
    def _maximization_step(source, target, estep_res, sigma2_p=None):
        return None

    def registration(self, target, w=0.0, maxiter=50, tol=0.001):
        """EM loop of the reference (bcpd.py:82-98) with the clouds, G and v_hat resident on the GPU."""
        assert self._tf_type is not None, "transformation type is None."
        target = _as_points(target)
        source = self._source
        assert source.ndim == 2 and target.ndim == 2, "source and target must have 2 dimensions."
        if source.shape[1] != target.shape[1] or source.shape[1] not in (2, 3):
            raise ValueError("source and target must both be (n, 2) or (n, 3) arrays.")
        res = self._initialize(target)
        plan = self._plan
        dim = source.shape[1]
        need_rmse = tol >= 0 or log.isEnabledFor(10)
        rmse = None
        for i in range(maxiter):
            rigid = res.transformation.rigid_trans
            # device transform z = s R (y' + v_hat) + t' in the centred frame: y' = y - cy, z' = z - cx
            t_c = np.asarray(rigid.t) + rigid.scale * np.dot(rigid.rot, self._cy) - self._cx
            p = np.zeros(_lib.PRG_NPARAMS)
            p[:13] = _params_block(rigid.rot, t_c, rigid.scale, dim)[:13]
            p[13] = res.sigma2
            plan.set_params(p)
            nu_d, nu, px = _estep_on_plan(plan, target.shape[0], dim, rigid.scale, res.alpha,
                                          _sigma_diag(res.sigma_mat, source.shape[0]), res.sigma2, w)
            px = px + np.outer(nu, self._cx)
            estep_res = EstepResult(nu_d, nu, float(np.sum(nu)), px, px / np.maximum(nu, _TINY)[:, None])
            t_source = res.transformation.transform(source) if need_rmse else None
            res = self._device_mstep(target, rigid, estep_res, res.sigma2)
            for c in self._callbacks:
                c(res.transformation)
            if need_rmse:
                tmp_rmse = mu.compute_rmse(t_source, target)
                log.debug("Iteration: {}, Criteria: {}".format(i, tmp_rmse))
                if rmse is not None and abs(rmse - tmp_rmse) < tol:
                    break
                rmse = tmp_rmse
        return res.transformation


class CombinedBCPD(BayesianCoherentPointDrift):
    """BCPD with similarity + non-rigid displacement (reference bcpd.py:101-151).

    lmd   : weight of the motion-coherence prior (multiplies G^-1)
    k     : Dirichlet concentration of the mixing weights (1e20 = effectively uniform alpha)
    gamma : scale of the initial sigma2
    """

    def __init__(self, source=None, lmd=2.0, k=1.0e20, gamma=1.0, device=None):
        super(CombinedBCPD, self).__init__(source, device)
        self._tf_type = tf.CombinedTransformation
        self.lmd = lmd
        self.k = k
        self.gamma = gamma

    def _ensure_plan(self):
        """Plan with the (centred) source and its inverse-multiquadric kernel on the GPU."""
        if self._plan is None:
            self._cy = self._source.mean(axis=0)
            plan = CpdPlan(self._device)
            plan.set_source(self._source - self._cy)
            plan.bcpd_build_g(1.0)  # bcpd.py:107 - mu.inverse_multiquadric_kernel(source, source), c = 1
            self._plan = plan
        return self._plan

    def _initialize(self, target):
        m, dim = self._source.shape
        plan = self._ensure_plan()
        self._cx = target.mean(axis=0)
        plan.set_target(target - self._cx)
        plan.set_w(np.zeros((m, dim)))  # v_hat = 0 for a fresh registration (G and the solve workspace stay as built)
        sigma2 = self.gamma * mu.squared_kernel_sum(self._source, target)
        return MstepResult(self._tf_type(np.identity(dim), np.zeros(dim)), None, np.ones(m), 1.0 / m, sigma2)

    def maximization_step(self, target, rigid_trans, estep_res, sigma2_p=None):
        """Reference signature (bcpd.py:113-116).  Uses ``estep_res.nu`` as given (it need not come from this
        object's last E-step)."""
        self._ensure_plan()
        return self._device_mstep(_as_points(target), rigid_trans, estep_res, sigma2_p, nu=estep_res.nu)

    @staticmethod
    def _maximization_step(source, target, rigid_trans, estep_res, gmat_inv, lmd, k, sigma2_p=None):
        raise NotImplementedError("the explicit-G^-1 M-step of the reference has no GPU counterpart: use "
                                  "CombinedBCPD.maximization_step (it never forms G^-1)")

    def _device_mstep(self, target, rigid_trans, estep_res, sigma2_p, nu=None):
        source = self._source
        nu_d, nu_v, n_p, px, x_hat = estep_res
        m, dim = source.shape
        s2s2 = rigid_trans.scale ** 2 / (sigma2_p ** 2)
        residual = rigid_trans.inverse().transform(x_hat) - source
        # v_hat = s2s2 Sigma diag(nu) residual and diag(Sigma), Sigma = (lmd G^-1 + s2s2 diag(nu))^-1   (bcpd.py:123-129)
        v_hat, sigma_diag = self._plan.bcpd_solve(self.lmd, s2s2, residual, nu)
        u_hat = source + v_hat
        # mixing weights (bcpd.py:130): alpha_m = exp(psi(k + nu_m) - psi(k M + N_p))
        alpha = np.exp(spsp.psi(self.k + nu_v) - spsp.psi(self.k * m + n_p))
        # similarity from the nu-weighted first and second moments of (x_hat, u_hat)   (bcpd.py:131-143)
        wts = nu_v / n_p
        x_m, u_m = wts @ x_hat, wts @ u_hat
        sigma2_m = float(wts @ sigma_diag)
        xc, uc = x_hat - x_m, u_hat - u_m
        s_xu = (xc * wts[:, None]).T @ uc
        s_uu = (uc * wts[:, None]).T @ uc + sigma2_m * np.identity(dim)
        left, _, right_t = np.linalg.svd(s_xu, full_matrices=True)
        flip = np.ones(dim)
        flip[-1] = np.linalg.det(left @ right_t)          # keep det(rot) = +1
        rot = (left * flip) @ right_t
        scale = np.trace(rot @ s_xu) / np.trace(s_uu)      # (sic) rot, not rot.T - as the reference, bcpd.py:141
        t = x_m - scale * (rot @ u_m)
        # residual variance against the PREVIOUS similarity applied to y + v_hat (sic, bcpd.py:145-150)
        y_hat = rigid_trans.transform(u_hat)
        s1 = nu_d @ np.einsum("nd,nd->n", target, target)
        s2 = np.einsum("md,md->", px, y_hat)
        s3 = nu_v @ np.einsum("md,md->m", y_hat, y_hat)
        sigma2 = (s1 - 2.0 * s2 + s3) / (n_p * dim) + scale ** 2 * sigma2_m
        return MstepResult(tf.CombinedTransformation(rot, t, scale, v_hat), u_hat, sigma_diag, alpha, sigma2)


def registration_bcpd(source, target, w=0.0, maxiter=50, tol=0.001, callbacks=(), **kwargs):
    """One-call BCPD with the reference's signature (bcpd.py:154-185): returns the CombinedTransformation.

    source, target : (n, 2|3) arrays or Open3D point clouds;  w : outlier mass in [0, 1)
    maxiter, tol   : stop after maxiter iterations or when the mean nearest-neighbour distance changes by < tol
    **kwargs       : ``lmd``, ``k``, ``gamma`` of :class:`CombinedBCPD` (and ``device``)
    """
    bcpd = CombinedBCPD(_as_points(source), **kwargs)
    bcpd.set_callbacks(list(callbacks))
    try:
        return bcpd.registration(_as_points(target), w, maxiter, tol)
    finally:
        bcpd._close_plan()
