"""numpy restatement of probreg's FilterReg rigid point-to-point iteration (TEST INFRASTRUCTURE).

Restates (paths relative to /root/reference):
  probreg/filterreg.py:78-108   FilterReg.expectation_step     -> ``expectation_step``
  probreg/filterreg.py:158-196  RigidFilterReg._maximization_step (pt2pt branch) -> ``maximization_step``
  probreg/filterreg.py:120-147  FilterReg.registration         -> ``registration``
  probreg/cc/kabsch.cc:6-56     computeKabsch (float32)        -> ``kabsch_f32``
  probreg/cc/kabsch.cc:58-109   computeKabsch2d                -> ``kabsch2d_f32``
on top of the C lattice restatement (oracle/permutohedral_oracle.c, bit-identical to the vendored
reference lattice).  numpy dtypes follow the reference's expressions (the filter returns float32,
the transform is float64, sigma2 starts as a float32 scalar) so that both run the same arithmetic.

Parity status: PINNED - checked against the unmodified reference filterreg.py executed through
oracle/ref_import.py (tests/golden/make_golden.py, tests/test_oracle_filterreg.py).
"""
from collections import namedtuple

import numpy as np

from . import permutohedral as ph

EstepResult = namedtuple("EstepResult", ["m0", "m1", "m2", "nx"])
MstepResult = namedtuple("MstepResult", ["rot", "t", "sigma2", "q"])


# ----------------------------------------------------------------------------------------------
# weighted Kabsch, float32 like the reference's Eigen build (types.h:19: Float = float)
# ----------------------------------------------------------------------------------------------
def _seq_sum32(a):
    """Sequential float32 accumulation over axis 0 (the reference's plain for-loop, kabsch.cc:16-21)."""
    return np.cumsum(a, axis=0, dtype=np.float32)[-1]


def kabsch_f32(model, target, weight):
    """kabsch.cc:6-56.  Centroids use ``w``, the cross-covariance uses ``w**2`` (:37-41)."""
    m = np.ascontiguousarray(model, dtype=np.float32)
    t = np.ascontiguousarray(target, dtype=np.float32)
    w = np.ascontiguousarray(weight, dtype=np.float32)
    total = _seq_sum32(w)
    if total == 0:
        return np.identity(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
    inv = np.float32(1.0) / total
    mc = _seq_sum32(w[:, None] * m) * inv
    tc = _seq_sum32(w[:, None] * t) * inv
    w2 = w * w
    cm = m - mc
    ct = t - tc
    terms = (w2[:, None] * cm)[:, :, None] * ct[:, None, :]
    hh = _seq_sum32(terms.reshape(-1, 9)).reshape(3, 3)
    hh = hh / _seq_sum32(w2)
    u, _, vt = np.linalg.svd(hh.astype(np.float32))
    v = vt.T
    ss = np.ones(3, dtype=np.float32)
    ss[2] = np.linalg.det(u @ v)
    r = (v * ss) @ u.T
    trans = tc - r @ mc
    return r.astype(np.float32), trans.astype(np.float32)


def kabsch2d_f32(model, target, weight):
    """kabsch.cc:58-109 (closed-form 2-D rotation through atan2, :98-102)."""
    m = np.ascontiguousarray(model, dtype=np.float32)
    t = np.ascontiguousarray(target, dtype=np.float32)
    w = np.ascontiguousarray(weight, dtype=np.float32)
    total = _seq_sum32(w)
    if total == 0:
        return np.identity(2, dtype=np.float32), np.zeros(2, dtype=np.float32)
    inv = np.float32(1.0) / total
    mc = _seq_sum32(w[:, None] * m) * inv
    tc = _seq_sum32(w[:, None] * t) * inv
    w2 = w * w
    terms = (w2[:, None] * (m - mc))[:, :, None] * (t - tc)[:, None, :]
    hh = _seq_sum32(terms.reshape(-1, 4)).reshape(2, 2) / _seq_sum32(w2)
    ang = np.arctan2(hh[0, 1] - hh[1, 0], hh[0, 0] + hh[1, 1])
    r = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]], dtype=np.float32)
    return r, (tc - r @ mc).astype(np.float32)


def pt2pl_f32(model, target, normal, weight):
    """cc/point_to_plane.cc:6-32 (computeTwistForPointToPlane), float32 like the reference's Eigen build.

    Normal equations of sum_k w_k (n_k . (t_k - v_k) - jac_k . tw)^2 with jac_k = [v_k x n_k, n_k]:
    returns (twist[6], r_sum) with r_sum = sum_k w_k^2 residual_k^2 (note the squared weight there, :26).
    """
    v = np.ascontiguousarray(model, dtype=np.float32)
    t = np.ascontiguousarray(target, dtype=np.float32)
    n = np.ascontiguousarray(normal, dtype=np.float32)
    w = np.ascontiguousarray(weight, dtype=np.float32)
    residual = np.einsum("kd,kd->k", n, t - v, dtype=np.float32)
    jac = np.concatenate([np.cross(v, n).astype(np.float32), n], axis=1)
    ata = np.einsum("k,ki,kj->ij", w, jac, jac, dtype=np.float32)
    atb = np.einsum("k,k,ki->i", w, residual, jac, dtype=np.float32)
    r_sum = np.sum(w * w * residual * residual, dtype=np.float32)
    tw = np.linalg.solve(ata.astype(np.float32), atb.astype(np.float32))  # selfadjointView<Upper>().ldlt().solve
    return tw.astype(np.float32), np.float32(r_sum)


def twist_trans(tw):
    """se3_op.py:21-41 (non-linear branch): Rodrigues rotation of the first three entries, translation = last three."""
    twd = np.linalg.norm(tw[:3])
    if twd == 0.0:
        return np.identity(3), tw[3:]
    ntw = tw[:3] / twd
    c, s = np.cos(twd), np.sin(twd)
    skew = np.array([[0.0, -ntw[2], ntw[1]], [ntw[2], 0.0, -ntw[0]], [-ntw[1], ntw[0], 0.0]])
    return c * np.identity(3) + (1.0 - c) * np.outer(ntw, ntw) + s * skew, tw[3:]


def twist_mul(tw, rot, t):
    """se3_op.py:44-56."""
    tr, tt = twist_trans(tw)
    return np.dot(tr, rot), np.dot(t, tr.T) + tt


# ----------------------------------------------------------------------------------------------
# E-step / M-step
# ----------------------------------------------------------------------------------------------
def expectation_step(t_source, target, y, sigma2, update_sigma2, alpha=0.015, prefer_ref=False, info=None,
                     target_normals=None):
    """filterreg.py:78-108: lattice over [t_source; target]/sigma, three filters, first m rows kept."""
    assert t_source.ndim == 2 and target.ndim == 2, "source and target must have 2 dimensions."
    m = t_source.shape[0]
    n = target.shape[0]
    sigma = np.sqrt(sigma2)
    fin = np.r_[t_source / sigma, target / sigma]
    lat = ph.Lattice(fin, True, prefer_ref=prefer_ref)
    with_blur = True
    if lat.lattice_size > n * alpha:  # :90-91 rebuild without the blur stage
        lat = ph.Lattice(fin, False, prefer_ref=prefer_ref)
        with_blur = False
    if info is not None:
        info.append((lat.lattice_size, with_blur))
    zero_m1 = np.zeros((m, 1))
    m0 = lat.filter(np.r_[zero_m1, np.ones((n, 1))]).flatten()[:m]
    m1 = lat.filter(np.r_[np.zeros((m, y.shape[1])), y])[:m]
    m2 = None
    if update_sigma2:
        m2 = lat.filter(np.r_[zero_m1, np.square(y).sum(axis=1)[:, None]]).flatten()[:m]
    nx = None
    if target_normals is not None:  # objective_type == 'pt2pl', filterreg.py:103-105
        nx = lat.filter(np.r_[np.zeros((m, y.shape[1])), target_normals])[:m]
    return EstepResult(m0, m1, m2, nx)


def maximization_step(t_source, target, es, rot_p, t_p, sigma2, w=0.0, objective_type="pt2pt"):
    """filterreg.py:158-182 + :190-196 (pt2pt).  Returns MstepResult(rot, t, sigma2, q); q None if all m0 == 0."""
    m, dim = t_source.shape
    n = target.shape[0]
    assert dim == 2 or dim == 3, "dim must be 2 or 3."
    m0, m1, m2, nx = es
    c = w / (1.0 - w) * n / m * (2.0 * sigma2 * np.pi) ** (dim / 2.0)
    nz = m0 != 0
    if not nz.any():
        return MstepResult(rot_p, t_p, sigma2, None)
    m0 = m0[nz]
    m1 = m1[nz]
    ts = t_source[nz]
    m1m0 = np.divide(m1.T, m0).T
    m0m0 = m0 / (m0 + c)
    drxdx = np.sqrt(m0m0 * 1.0 / sigma2)
    if objective_type == "pt2pl":  # filterreg.py:183-186
        nxm0 = (nx[nz].T / m0).T
        tw, q = pt2pl_f32(ts, m1m0, nxm0, drxdx)
        rot, t = twist_mul(tw, rot_p, t_p)
    else:
        if dim == 2:
            dr, dt = kabsch2d_f32(ts, m1m0, drxdx)
        else:
            dr, dt = kabsch_f32(ts, m1m0, drxdx)
        rx = np.multiply(drxdx, (ts - m1m0).T).T
        rot, t = np.dot(dr, rot_p), np.dot(t_p, dr.T) + dt
        q = np.linalg.norm(rx, ord=2, axis=1).sum()
    if m2 is not None:
        m2 = m2[nz]
        sigma2 = ((m0 * np.square(ts).sum(axis=1) - 2.0 * (ts * m1).sum(axis=1) + m2) / (m0 + c)).sum()
        sigma2 /= 3.0 * m0m0.sum()  # the reference hard-codes 3.0 here (:195), also for 2-D data
    return MstepResult(rot, t, sigma2, q)


def squared_kernel_sum_f32(x, y):
    """mu.squared_kernel_sum as the reference evaluates it (float32 matrix, float32 scalar result)."""
    from . import cpd_numpy as co

    return np.float32(co.squared_kernel_sum(x, y))


def registration(source, target, sigma2=None, update_sigma2=False, w=0.0, maxiter=50, tol=0.001, min_sigma2=1.0e-4,
                 rot0=None, t0=None, prefer_ref=False, history=None, info=None, target_normals=None,
                 objective_type="pt2pt", feature_fn=None):
    """filterreg.py:120-147 (``feature_fn`` None = the identity).  Returns (rot, t, sigma2_returned, q, n_iter)."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    dim = source.shape[1]
    rot = np.identity(dim) if rot0 is None else np.asarray(rot0, dtype=np.float64)
    t = np.zeros(dim) if t0 is None else np.asarray(t0, dtype=np.float64)
    q = None
    feat = (lambda x: x) if feature_fn is None else feature_fn
    ftarget = feat(target)  # filterreg.py:125
    if sigma2 is None:
        sigma2 = max(squared_kernel_sum_f32(feat(source), ftarget), min_sigma2)
    res = None
    n_iter = 0
    for _ in range(maxiter):
        ts = np.dot(source, rot.T) + t  # RigidTransformation._transform with scale 1 (transformation.py:49-50)
        es = expectation_step(feat(ts), ftarget, target, sigma2, update_sigma2, prefer_ref=prefer_ref, info=info,
                              target_normals=target_normals if objective_type == "pt2pl" else None)
        res = maximization_step(ts, target, es, rot, t, sigma2, w=w, objective_type=objective_type)
        if res.q is None:
            res = res._replace(q=q)
            break
        rot, t = res.rot, res.t
        sigma2 = max(res.sigma2, min_sigma2)
        n_iter += 1
        if history is not None:
            history.append((res.sigma2, res.q))
        if q is not None and abs(res.q - q) < tol:
            break
        q = res.q
    return res.rot, res.t, res.sigma2, res.q, n_iter
