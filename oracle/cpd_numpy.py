"""numpy fp64 restatement of probreg's CPD EM iteration (TEST INFRASTRUCTURE).

Every function cites the reference lines it restates (paths relative to
/root/reference).  The E-step is evaluated in row blocks of the source so that
no M x N temporary larger than ``chunk`` x N is ever allocated; apart from the
order of the fp64 summations it is the same arithmetic as the reference, and
``tests/test_oracle_golden.py`` checks it against fixtures produced by the
reference's own code (``tests/golden/make_golden.py``).

Parity status: PINNED against the reference executed in the build container.
"""
from collections import namedtuple

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)

EstepResult = namedtuple("EstepResult", ["pt1", "p1", "px", "n_p"])
MstepResult = namedtuple("MstepResult", ["params", "sigma2", "q"])


# ----------------------------------------------------------------------------------------------
# one-off quantities
# ----------------------------------------------------------------------------------------------
def squared_kernel_sum(x, y, chunk=None):
    """sigma^2 initialiser: math_utils.py:28-29 over cc/math_utils.cc:5-15.

    The reference builds the dense M x N matrix of squared distances in float32
    (Eigen ``MatrixXf``), then ``ndarray.sum()`` (float32 pairwise summation) and
    divides by M*D*N.  Restated block-wise: the float32 squared distances are
    summed per block in float32 (numpy pairwise) and the block sums are combined
    in float64, which differs from a single float32 ``.sum()`` only in the last
    float32 digit of the total.
    """
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    y32 = np.ascontiguousarray(y, dtype=np.float32)
    if chunk is None:
        # one block (= the reference's single float32 .sum(), bit for bit) as long as the temporary stays < 1 GB
        chunk = max(1, min(x32.shape[0], (1 << 28) // max(1, y32.shape[0] * x32.shape[1])))
    total = 0.0
    for s in range(0, x32.shape[0], chunk):
        d = x32[s:s + chunk, None, :] - y32[None, :, :]
        total += float(np.einsum("mnd,mnd->mn", d, d, dtype=np.float32).sum(dtype=np.float32))
    return total / (x32.shape[0] * x32.shape[1] * y32.shape[0])


def squared_kernel_sum_closed_form(x, y):
    """Same quantity through the O(M+N) identity, all in fp64."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    m, d = x.shape
    n = y.shape[0]
    # [r6] both clouds are shifted by ONE common point first (the quantity is a sum of squared DIFFERENCES: translation does not
    # change it): with clouds ~1000 units from the origin the three terms are ~1e15 and their difference ~1e9 - the un-shifted
    # identity lost 4e-7 there (tools/far_offset_bisect.py; the reference itself sums the differences, math_utils.py:28-29),
    # and ten EM iterations carried that to 5e-6 in sigma2: the round-5 fuzz's two worst cases were the oracle's, not the GPU's
    c = (x.sum(axis=0) + y.sum(axis=0)) / (m + n)
    x = x - c
    y = y - c
    return (n * np.sum(x * x) + m * np.sum(y * y) - 2.0 * np.dot(x.sum(axis=0), y.sum(axis=0))) / (m * d * n)


def rbf_kernel(x, y, beta, chunk=2048):
    """G matrix: transformation.py:91-99 -> cc/math_utils.cc:17-19 (float32, exp(-d2/(2*beta)))."""
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    y32 = np.ascontiguousarray(y, dtype=np.float32)
    g = np.empty((x32.shape[0], y32.shape[0]), dtype=np.float32)
    for s in range(0, x32.shape[0], chunk):
        d = x32[s:s + chunk, None, :] - y32[None, :, :]
        d2 = np.einsum("mnd,mnd->mn", d, d, dtype=np.float32)
        g[s:s + chunk] = np.exp(-d2 / np.float32(2.0 * beta))
    return g


# ----------------------------------------------------------------------------------------------
# E-step
# ----------------------------------------------------------------------------------------------
def _sqdist(a, b):
    # scipy cdist(.., 'sqeuclidean') evaluates sum_k (a_k - b_k)^2 in float64 (cpd.py:74).
    d = a[:, None, :] - b[None, :, :]
    return np.einsum("mnd,mnd->mn", d, d)


def expectation_step(t_source, target, sigma2, w=0.0, chunk=1024):
    """cpd.py:71-88 in two sweeps over row blocks of ``t_source``.

    Sweep 1 accumulates the column sums ``den`` (cpd.py:80), applies the
    ``den == 0 -> eps32`` rule (:81) and adds the uniform term ``c`` (:78-79,:82).
    Sweep 2 forms P = K / den per block and accumulates pt1, p1 and P.X (:84-87).
    """
    ts = np.asarray(t_source, dtype=np.float64)
    x = np.asarray(target, dtype=np.float64)
    assert ts.ndim == 2 and x.ndim == 2, "source and target must have 2 dimensions."
    m, dim = ts.shape
    n = x.shape[0]
    inv = -1.0 / (2.0 * sigma2)
    c = (2.0 * np.pi * sigma2) ** (dim * 0.5)
    c *= w / (1.0 - w) * m / n
    den = np.zeros(n)
    for s in range(0, m, chunk):
        den += np.exp(_sqdist(ts[s:s + chunk], x) * inv).sum(axis=0)
    den[den == 0] = EPS32
    den += c
    pt1 = np.zeros(n)
    p1 = np.empty(m)
    px = np.empty((m, dim))
    for s in range(0, m, chunk):
        p = np.exp(_sqdist(ts[s:s + chunk], x) * inv)
        p /= den
        pt1 += p.sum(axis=0)
        p1[s:s + chunk] = p.sum(axis=1)
        px[s:s + chunk] = p @ x
    return EstepResult(pt1, p1, px, float(np.sum(p1)))


def expectation_step_unchunked(t_source, target, sigma2, w=0.0):
    """cpd.py:71-88 line by line, the reference's OWN formulation: one dense M x N float64 matrix through scipy's
    ``cdist`` (the reference's ``distance_module``, cpd.py:57), ``exp``, column sums, ``divide``, row / column sums and
    ``dot`` - at least nine full passes over the matrix.  This is what `cpu_baseline.reference_numpy` times on the bench
    host (the reference tree itself does not travel to the GPU box); pinned to the reference's outputs at 1e-12 by
    tests/test_oracle_golden.py beside the chunked form."""
    from scipy.spatial import distance as scipy_distance

    t_source = np.asarray(t_source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    assert t_source.ndim == 2 and target.ndim == 2, "source and target must have 2 dimensions."
    pmat = scipy_distance.cdist(t_source, target, "sqeuclidean")          # :74
    pmat = np.exp(-pmat / (2.0 * sigma2))                                  # :76
    c = (2.0 * np.pi * sigma2) ** (t_source.shape[1] * 0.5)                # :78
    c *= w / (1.0 - w) * t_source.shape[0] / target.shape[0]               # :79
    den = np.sum(pmat, axis=0)                                             # :80
    den[den == 0] = np.finfo(np.float32).eps                               # :81
    den += c                                                               # :82
    pmat = np.divide(pmat, den)                                            # :84
    pt1 = np.sum(pmat, axis=0)                                             # :85
    p1 = np.sum(pmat, axis=1)                                              # :86
    px = np.dot(pmat, target)                                              # :87
    return EstepResult(pt1, p1, px, float(np.sum(p1)))                     # :88


# ----------------------------------------------------------------------------------------------
# M-steps
# ----------------------------------------------------------------------------------------------
def _common_moments(source, target, es):
    # cpd.py:169-175 / 227-233
    pt1, p1, px, n_p = es
    mu_x = px.sum(axis=0) / n_p
    mu_y = source.T @ p1 / n_p
    target_hat = target - mu_x
    source_hat = source - mu_y
    a = px.T @ source_hat - np.outer(mu_x, p1 @ source_hat)
    return mu_x, mu_y, target_hat, source_hat, a


def mstep_rigid(source, target, es, update_scale=True):
    """cpd.py:160-192.  Returns params dict(rot, t, scale)."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    pt1, p1, px, n_p = es
    dim = source.shape[1]
    mu_x, mu_y, target_hat, source_hat, a = _common_moments(source, target, es)
    u, _, vh = np.linalg.svd(a, full_matrices=True)
    cdiag = np.ones(dim)
    cdiag[-1] = np.linalg.det(u @ vh)
    rot = (u * cdiag) @ vh
    tr_atr = np.trace(a.T @ rot)
    tr_yp1y = np.trace((source_hat.T * p1) @ source_hat)
    scale = tr_atr / tr_yp1y if update_scale else 1.0
    t = mu_x - scale * rot @ mu_y
    tr_xp1x = np.trace((target_hat.T * pt1) @ target_hat)
    if update_scale:
        sigma2 = (tr_xp1x - scale * tr_atr) / (n_p * dim)
    else:
        # NB the reference's no-scale branch is "+ tr_yp1y - scale*tr_atr" (cpd.py:188), kept as is.
        sigma2 = (tr_xp1x + tr_yp1y - scale * tr_atr) / (n_p * dim)
    sigma2 = max(sigma2, EPS32)
    q = (tr_xp1x - 2.0 * scale * tr_atr + scale ** 2 * tr_yp1y) / (2.0 * sigma2)
    q += dim * n_p * 0.5 * np.log(sigma2)
    return MstepResult(dict(rot=rot, t=t, scale=scale), sigma2, q)


def mstep_affine(source, target, es):
    """cpd.py:219-244.  Returns params dict(b, t)."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    pt1, p1, px, n_p = es
    dim = source.shape[1]
    mu_x, mu_y, target_hat, source_hat, a = _common_moments(source, target, es)
    yp1y = (source_hat.T * p1) @ source_hat
    b = np.linalg.solve(yp1y.T, a.T).T
    t = mu_x - b @ mu_y
    tr_xp1x = np.trace((target_hat.T * pt1) @ target_hat)
    tr_ab = np.trace(a @ b.T)
    sigma2 = (tr_xp1x - tr_ab) / (n_p * dim)
    sigma2 = max(sigma2, EPS32)
    q = (tr_xp1x - 2.0 * tr_ab + tr_ab) / (2.0 * sigma2)
    q += dim * n_p * 0.5 * np.log(sigma2)
    return MstepResult(dict(b=b, t=t), sigma2, q)


def mstep_nonrigid(source, target, es, sigma2_p, g, lmd):
    """cpd.py:284-303.  ``g`` is the float32 G matrix; the products are evaluated in float64."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    pt1, p1, px, n_p = es
    m, dim = source.shape
    lhs = (p1 * g).T + lmd * sigma2_p * np.identity(m)
    rhs = px - (source.T * p1).T
    w = np.linalg.solve(lhs, rhs)
    t = source + g @ w
    tr_xp1x = np.trace((target.T * pt1) @ target)
    tr_pxt = np.trace(px.T @ t)
    tr_tpt = np.trace((t.T * p1) @ t)
    sigma2 = (tr_xp1x - 2.0 * tr_pxt + tr_tpt) / (n_p * dim)
    return MstepResult(dict(w=w), sigma2, sigma2)


def mstep_nonrigid_constrained(source, target, es, sigma2_p, g, lmd, alpha, p1_tilde, px_tilde):
    """ConstrainedNonRigidCPD._maximization_step, cpd.py:377-404."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    pt1, p1, px, n_p = es
    m, dim = source.shape
    lhs = (p1 * g).T + sigma2_p / alpha * (p1_tilde * g).T + lmd * sigma2_p * np.identity(m)
    rhs = px - (source.T * p1).T + sigma2_p / alpha * (px_tilde - (source.T * p1_tilde).T)
    w = np.linalg.solve(lhs, rhs)
    t = source + g @ w
    tr_xp1x = np.trace((target.T * pt1) @ target)
    tr_pxt = np.trace(px.T @ t)
    tr_tpt = np.trace((t.T * p1) @ t)
    sigma2 = (tr_xp1x - 2.0 * tr_pxt + tr_tpt) / (n_p * dim)
    return MstepResult(dict(w=w), sigma2, sigma2)


# ----------------------------------------------------------------------------------------------
# transforms + driver
# ----------------------------------------------------------------------------------------------
def transform(kind, params, source, g=None):
    """transformation.py:49-50 (rigid), :77-78 (affine), :101-102 (nonrigid)."""
    source = np.asarray(source, dtype=np.float64)
    if kind == "rigid":
        return params["scale"] * source @ params["rot"].T + params["t"]
    if kind == "affine":
        return source @ params["b"].T + params["t"]
    if kind in ("nonrigid", "nonrigid_constrained"):
        return source + g @ params["w"]
    raise ValueError("Unknown transformation type %s" % kind)


def registration(kind, source, target, w=0.0, maxiter=50, tol=0.001, update_scale=True,
                 tf_init_params=None, beta=2.0, lmd=2.0, chunk=1024, closed_form_init=False,
                 history=None, alpha=1e-8, idx_source=None, idx_target=None, c_estep=False):
    """EM driver, cpd.py:106-120 with the per-type ``_initialize`` (:145-153, :209-217, :277-282).

    Returns (params, sigma2, q, n_iter).  ``history`` (a list) receives (sigma2, q) per iteration.
    ``c_estep``: the E-step through oracle/cpd_estep_c.c (C / OpenMP fp64, the same arithmetic, held to this module's
    ``expectation_step`` by tests/test_oracle_c.py) instead of the chunked numpy one - an order of magnitude faster.
    """
    estep = expectation_step
    if c_estep:
        from . import cpd_c

        def estep(ts, target, sigma2, w, chunk=None):
            return EstepResult(*cpd_c.expectation_step(ts, target, sigma2, w))
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    dim = source.shape[1]
    sks = squared_kernel_sum_closed_form if closed_form_init else squared_kernel_sum
    sigma2 = sks(source, target)
    q = 1.0 + target.shape[0] * dim * 0.5 * np.log(sigma2)
    g = None
    if kind == "rigid":
        params = dict(rot=np.identity(dim), t=np.zeros(dim), scale=1.0)
        if tf_init_params:
            params.update(tf_init_params)
    elif kind == "affine":
        params = dict(b=np.identity(dim), t=np.zeros(dim))
        if tf_init_params:
            params.update(tf_init_params)
    elif kind in ("nonrigid", "nonrigid_constrained"):
        g = rbf_kernel(source, source, beta)
        params = dict(w=np.zeros_like(source))
        if kind == "nonrigid_constrained":  # cpd.py:370-376
            p_tilde = np.zeros((source.shape[0], target.shape[0]))
            if idx_source is not None and idx_target is not None:
                p_tilde[idx_source, idx_target] = 1
            p1_tilde = p_tilde.sum(axis=1)
            px_tilde = p_tilde @ target
    else:
        raise ValueError("Unknown transformation type %s" % kind)
    n_iter = 0
    for _ in range(maxiter):
        ts = transform(kind, params, source, g)
        es = estep(ts, target, sigma2, w, chunk=chunk)
        if kind == "rigid":
            params, sigma2_new, q_new = mstep_rigid(source, target, es, update_scale)
        elif kind == "affine":
            params, sigma2_new, q_new = mstep_affine(source, target, es)
        elif kind == "nonrigid_constrained":
            params, sigma2_new, q_new = mstep_nonrigid_constrained(source, target, es, sigma2, g, lmd, alpha,
                                                                   p1_tilde, px_tilde)
        else:
            params, sigma2_new, q_new = mstep_nonrigid(source, target, es, sigma2, g, lmd)
        sigma2 = sigma2_new
        n_iter += 1
        if history is not None:
            history.append((sigma2, q_new))
        if abs(q_new - q) < tol:
            q = q_new
            break
        q = q_new
    return params, sigma2, q, n_iter


# ----------------------------------------------------------------------------------------------
# moment form of the rigid / affine M-step - used by the multi-process
# tests to show that the 23 moments are shard-additive and sufficient
# ----------------------------------------------------------------------------------------------
def moments_from_estep(source, target, es):
    """The 32-double MOMENTS block of include/probreg_hip.h from an EstepResult (3-D padded)."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    pt1, p1, px, _ = es
    dim = source.shape[1]
    y = np.zeros((source.shape[0], 3))
    y[:, :dim] = source
    pxx = np.zeros((source.shape[0], 3))
    pxx[:, :dim] = px
    mom = np.zeros(32)
    mom[0] = p1.sum()
    mom[1:4] = pxx.sum(axis=0)
    mom[4:7] = y.T @ p1
    mom[7:16] = (pxx.T @ y).ravel()
    syy = (y.T * p1) @ y
    mom[16:22] = [syy[0, 0], syy[0, 1], syy[0, 2], syy[1, 1], syy[1, 2], syy[2, 2]]
    mom[22] = float(np.sum(pt1 * np.einsum("nd,nd->n", target, target)))
    return mom


def mstep_from_moments(kind, mom, dim, update_scale=True):
    """Rigid / affine M-step (cpd.py:160-192 / 219-244) written on the moment block."""
    s0 = mom[0]
    mu_x = mom[1:4] / s0
    mu_y = mom[4:7] / s0
    a = mom[7:16].reshape(3, 3) - np.outer(mom[1:4], mu_y)
    syy = np.array([[mom[16], mom[17], mom[18]], [mom[17], mom[19], mom[20]], [mom[18], mom[20], mom[21]]])
    ypy = syy - s0 * np.outer(mu_y, mu_y)
    a, ypy, mu_x, mu_y = a[:dim, :dim], ypy[:dim, :dim], mu_x[:dim], mu_y[:dim]
    tr_yp1y = np.trace(ypy)
    tr_xp1x = mom[22] - s0 * float(mu_x @ mu_x)
    if kind == "rigid":
        u, _, vh = np.linalg.svd(a, full_matrices=True)
        cdiag = np.ones(dim)
        cdiag[-1] = np.linalg.det(u @ vh)
        rot = (u * cdiag) @ vh
        tr_atr = np.trace(a.T @ rot)
        scale = tr_atr / tr_yp1y if update_scale else 1.0
        t = mu_x - scale * rot @ mu_y
        if update_scale:
            sigma2 = (tr_xp1x - scale * tr_atr) / (s0 * dim)
        else:
            sigma2 = (tr_xp1x + tr_yp1y - scale * tr_atr) / (s0 * dim)
        sigma2 = max(sigma2, EPS32)
        q = (tr_xp1x - 2.0 * scale * tr_atr + scale ** 2 * tr_yp1y) / (2.0 * sigma2) + dim * s0 * 0.5 * np.log(sigma2)
        return MstepResult(dict(rot=rot, t=t, scale=scale), sigma2, q)
    b = np.linalg.solve(ypy.T, a.T).T
    t = mu_x - b @ mu_y
    tr_ab = np.trace(a @ b.T)
    sigma2 = max((tr_xp1x - tr_ab) / (s0 * dim), EPS32)
    q = (tr_xp1x - tr_ab) / (2.0 * sigma2) + dim * s0 * 0.5 * np.log(sigma2)
    return MstepResult(dict(b=b, t=t), sigma2, q)
