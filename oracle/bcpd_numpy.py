"""CPU restatement of the reference's BCPD (probreg/bcpd.py) in numpy float64.  TEST INFRASTRUCTURE ONLY.

Pinned against the unmodified reference through ``oracle/ref_import.py`` (tests/golden/make_golden.py bcpd ->
tests/golden/bcpd_golden.npz, checked by tests/test_oracle_bcpd.py).  Differences from the reference, all
deliberate and all about memory, not arithmetic:
  * the E-step (bcpd.py:53-72) is evaluated in row chunks and ``px`` as ``P @ target`` instead of through the
    (MD x ND) Kronecker product of :69-70 - same numbers, no 9x blow-up;
  * the M-step (bcpd.py:119-151) multiplies ``Sigma diag(nu) R`` directly instead of through ``np.kron`` (:126-128)
    and reads ``diag(Sigma)`` where the reference reads ``np.diag(sigma_mat)``;
  * ``gmat_inv`` is computed from the float32 kernel matrix either in float32, as the reference does
    (``np.linalg.inv`` of a float32 array, bcpd.py:108 - ``inv_dtype=np.float32``), or in float64
    (``inv_dtype=np.float64``): the two agree only while G is well conditioned, which is why the parity fixtures use
    well-separated points.
"""
from collections import namedtuple

import numpy as np
import scipy.special as spsp
from scipy.spatial import cKDTree

from . import cpd_numpy as co

EstepResult = namedtuple("EstepResult", ["nu_d", "nu", "n_p", "px", "x_hat"])
MstepResult = namedtuple("MstepResult", ["rot", "t", "scale", "v", "u_hat", "sigma_diag", "alpha", "sigma2"])


def inverse_multiquadric_kernel(x, y, c=1.0):
    """cc/math_utils.cc:32-34 in float32: 1 / sqrt(|x_i - y_j|^2 + c)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty((x.shape[0], y.shape[0]), dtype=np.float32)
    for i in range(y.shape[0]):
        d = x - y[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + (d[:, 2] * d[:, 2] if x.shape[1] > 2 else np.float32(0))
        out[:, i] = np.float32(1.0) / np.sqrt(d2 + np.float32(c))
    return out


def expectation_step(t_source, target, scale, alpha, sigma_diag, sigma2, w=0.0, chunk=512):
    """bcpd.py:53-72.  ``sigma_diag`` = np.diag(sigma_mat)."""
    t_source = np.asarray(t_source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    m, dim = t_source.shape
    n = target.shape[0]
    a = np.broadcast_to(np.asarray(alpha, dtype=np.float64), (m,)) * (1.0 - w)
    a = a * np.exp(-(scale ** 2) / (2.0 * sigma2) * np.asarray(sigma_diag, dtype=np.float64) * dim)
    a = a / (2.0 * np.pi * sigma2) ** (dim * 0.5)
    den = np.full(n, w / n)
    for s in range(0, m, chunk):                      # den_n = w/N + sum_m a_m exp(-d2 / 2 sigma2)   (:63)
        d2 = ((t_source[s:s + chunk, None, :] - target[None, :, :]) ** 2).sum(axis=2)
        den += (a[s:s + chunk, None] * np.exp(-d2 / (2.0 * sigma2))).sum(axis=0)
    den[den == 0] = np.finfo(np.float32).eps          # :64
    nu_d = np.zeros(n)
    nu = np.zeros(m)
    px = np.zeros((m, dim))
    for s in range(0, m, chunk):
        d2 = ((t_source[s:s + chunk, None, :] - target[None, :, :]) ** 2).sum(axis=2)
        p = a[s:s + chunk, None] * np.exp(-d2 / (2.0 * sigma2)) / den[None, :]
        nu_d += p.sum(axis=0)
        nu[s:s + chunk] = p.sum(axis=1)
        px[s:s + chunk] = p @ target
    with np.errstate(divide="ignore", invalid="ignore"):
        x_hat = px / nu[:, None]
    return EstepResult(nu_d, nu, float(nu.sum()), px, x_hat)


def _rigid(rot, t, scale, pts):
    return scale * np.dot(pts, rot.T) + t


def maximization_step(source, target, rot_p, t_p, scale_p, es, gmat_inv, lmd, k, sigma2_p):
    """bcpd.py:119-151 for the previous similarity (rot_p, t_p, scale_p)."""
    nu_d, nu, n_p, px, x_hat = es
    m, dim = source.shape
    s2s2 = scale_p ** 2 / (sigma2_p ** 2)
    sigma_mat = np.linalg.inv(lmd * gmat_inv + s2s2 * np.diag(nu))
    inv_rot, inv_scale = rot_p.T, 1.0 / scale_p
    inv_t = -np.dot(rot_p.T, t_p) / scale_p
    residual = _rigid(inv_rot, inv_t, inv_scale, x_hat) - source
    v_hat = s2s2 * (sigma_mat @ (nu[:, None] * residual))
    u_hat = source + v_hat
    alpha = np.exp(spsp.psi(k + nu) - spsp.psi(k * m + n_p))
    sig_d = np.diag(sigma_mat)
    x_m = np.sum(nu * x_hat.T, axis=1) / n_p
    sigma2_m = np.sum(nu * sig_d) / n_p
    u_m = np.sum(nu * u_hat.T, axis=1) / n_p
    u_hm = u_hat - u_m
    s_xu = np.matmul(np.multiply(nu, (x_hat - x_m).T), u_hm) / n_p
    s_uu = np.matmul(np.multiply(nu, u_hm.T), u_hm) / n_p + sigma2_m * np.identity(dim)
    phi, _, psih = np.linalg.svd(s_xu, full_matrices=True)
    c = np.ones(dim)
    c[-1] = np.linalg.det(np.dot(phi, psih))
    rot = np.matmul(phi * c, psih)
    scale = np.trace(np.matmul(rot, s_xu)) / np.trace(s_uu)
    t = x_m - scale * np.dot(rot, u_m)
    y_hat = _rigid(rot_p, t_p, scale_p, source + v_hat)
    s1 = np.dot(nu_d, np.sum(target * target, axis=1))
    s2 = np.sum(px * y_hat)
    s3 = np.dot(nu, np.sum(y_hat * y_hat, axis=1))
    sigma2 = (s1 - 2.0 * s2 + s3) / (n_p * dim) + scale ** 2 * sigma2_m
    return MstepResult(rot, t, scale, v_hat, u_hat, sig_d, alpha, sigma2)


def registration(source, target, w=0.0, maxiter=50, tol=0.001, lmd=2.0, k=1.0e20, gamma=1.0, inv_dtype=np.float32):
    """bcpd.py:82-98 + CombinedBCPD._initialize (:105-111).  Returns (MstepResult, iterations run)."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    m, dim = source.shape
    gmat = inverse_multiquadric_kernel(source, source)
    gmat_inv = np.linalg.inv(gmat.astype(inv_dtype))
    sigma2 = gamma * co.squared_kernel_sum(source, target)
    res = MstepResult(np.identity(dim), np.zeros(dim), 1.0, 0.0, None, np.ones(m), 1.0 / m, sigma2)
    tree = cKDTree(target, leafsize=10)
    rmse = None
    it = 0
    for it in range(1, maxiter + 1):
        t_source = _rigid(res.rot, res.t, res.scale, source + res.v)
        es = expectation_step(t_source, target, res.scale, res.alpha, res.sigma_diag, res.sigma2, w)
        res = maximization_step(source, target, res.rot, res.t, res.scale, es, gmat_inv, lmd, k, res.sigma2)
        tmp = sum(tree.query(t_source)[0]) / t_source.shape[0]   # math_utils.py:32-33
        if rmse is not None and abs(rmse - tmp) < tol:
            break
        rmse = tmp
    return res, it
