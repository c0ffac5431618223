"""The algebra of the fused single sweep (csrc/cpd.hip k_colfinal_fused / k_fused_final), in numpy fp64 on the
CPU: the 23 moments of the rigid M-step (cpd.py:160-192) taken from per-COLUMN sums over the TRANSFORMED source, relative to
per-block origins and with per-column exponent offsets, mapped back to the source's own frame - against the same moments taken
the reference's way (rows of P, cpd.py:84-88, 169-183), and the M-step that follows against the oracle's."""
import numpy as np
import pytest

from oracle import cpd_numpy as co


def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


@pytest.mark.parametrize("w,sigma2,scale", [(0.0, 0.05, 1.0), (0.2, 0.01, 1.07), (0.1, 0.3, 0.93)])
def test_column_side_sums_give_the_reference_moments(w, sigma2, scale):
    rng = np.random.default_rng(5)
    m, n, block = 700, 900, 128
    y = rng.normal(size=(m, 3))
    rot, t = _rot(0.3, -0.2, 0.5), np.array([0.3, -0.1, 0.2])
    x = (scale * 1.02) * y[rng.integers(0, m, n)] @ _rot(0.35, -0.25, 0.45).T + t + 0.05 * rng.normal(size=(n, 3))
    z = scale * y @ rot.T + t                                   # transformation.py:49-50
    # the reference's E-step and its row-side moments
    es = co.expectation_step(z, x, sigma2, w)
    pt1, p1, px, n_p = es
    ref = dict(S0=n_p, Sx=px.sum(0), Sy=y.T @ p1, Sxy=px.T @ y, trSyy=float(np.sum(p1 * np.sum(y * y, axis=1))),
               Sxx=float(np.sum(pt1 * np.sum(x * x, axis=1))))
    # the fused sweep's per-column sums: K = exp2(kk d^2 + L_n) with an arbitrary per-column offset, origin o per block of columns
    kk = -np.log2(np.e) / (2.0 * sigma2)
    c = (2.0 * np.pi * sigma2) ** 1.5 * w / (1.0 - w) * m / n if w > 0 else 0.0
    d2 = ((x[:, None, :] - z[None, :, :]) ** 2).sum(-1)         # [n][m]
    L = rng.uniform(-3.0, 40.0, n)                               # exponent offsets (the kernel: col_seed_offset)
    K = np.exp2(kk * d2 + L[:, None])
    mom = np.zeros(24)
    for b0 in range(0, n, block):
        sl = slice(b0, min(b0 + block, n))
        o = 0.5 * (x[sl].min(0) + x[sl].max(0))                  # the block's origin
        zr = z - o
        A = K[sl].sum(1)
        B = K[sl] @ zr
        E = K[sl] @ np.sum(zr * zr, axis=1)
        den = A * np.exp2(-L[sl])
        pd = den / (den + c)
        qn = pd / A
        pz = qn[:, None] * B + pd[:, None] * o
        mom[0] += pd.sum()
        mom[1:4] += (pd[:, None] * x[sl]).sum(0)
        mom[4:7] += pz.sum(0)
        mom[7:16] += (x[sl].T @ pz).ravel()
        mom[16] += np.sum(qn * (E + 2.0 * B @ o)) + np.sum(pd) * (o @ o)
        mom[22] += np.sum(pd * np.sum(x[sl] * x[sl], axis=1))
    # k_fused_final: z-side sums back to the source's frame through y = R^T (z - t) / s
    S0, Sx, Sz, Sxz = mom[0], mom[1:4], mom[4:7], mom[7:16].reshape(3, 3)
    Sy = rot.T @ (Sz - S0 * t) / scale
    Sxy = (Sxz - np.outer(Sx, t)) @ rot / scale
    trSyy = (mom[16] - 2.0 * t @ Sz + S0 * (t @ t)) / scale ** 2
    assert abs(S0 - ref["S0"]) < 1e-11 * ref["S0"]
    assert np.max(np.abs(Sx - ref["Sx"])) < 1e-10 * ref["S0"]
    assert np.max(np.abs(Sy - ref["Sy"])) < 1e-10 * ref["S0"]
    assert np.max(np.abs(Sxy - ref["Sxy"])) < 1e-10 * ref["S0"]
    assert abs(trSyy - ref["trSyy"]) < 1e-10 * ref["S0"]
    assert abs(mom[22] - ref["Sxx"]) < 1e-10 * ref["S0"]
    # ... and the rigid M-step from these moments (the arithmetic of k_mstep) is the oracle's
    mu_x, mu_y = Sx / S0, Sy / S0
    a = Sxy - np.outer(Sx, mu_y)
    u, _, vh = np.linalg.svd(a)
    cdiag = np.ones(3)
    cdiag[-1] = np.linalg.det(u @ vh)
    r_new = (u * cdiag) @ vh
    tr_atr = np.trace(a.T @ r_new)
    tr_yp1y = trSyy - S0 * (mu_y @ mu_y)
    s_new = tr_atr / tr_yp1y
    t_new = mu_x - s_new * r_new @ mu_y
    sigma2_new = (mom[22] - S0 * (mu_x @ mu_x) - s_new * tr_atr) / (S0 * 3)
    p, s2, _q = co.mstep_rigid(y, x, es)
    assert np.max(np.abs(r_new - p["rot"])) < 1e-9 and abs(s_new - p["scale"]) < 1e-9 and np.max(np.abs(t_new - p["t"])) < 1e-9
    assert abs(sigma2_new - s2) < 1e-8 * s2


def _col_offset(kk, dmin):
    """prg::col_offset (cpd_sweeps.h): the float just below -kk * dmin."""
    p = np.float32(-(np.float32(kk) * np.float32(dmin)))
    if not p > 0:
        return np.float32(0.0)
    return np.frombuffer(np.uint32(np.frombuffer(np.float32(p).tobytes(), np.uint32)[0] - 1).tobytes(), np.float32)[0]


@pytest.mark.parametrize("w,sigma2,scale,nseg", [(0.0, 0.02, 1.0, 1), (0.15, 0.004, 1.05, 5), (0.1, 0.2, 0.95, 3)])
def test_residual_column_sums_give_the_reference_moments(w, sigma2, scale, nseg):
    """DESIGN.md 3.1f (k_colpass_cull<true> / k_colpass_queue<true> -> k_colfinal_resid -> k_fused_final): per column and per
    segment of the source (min d^2, A, U, R) with A = sum K, U = sum K (x - z), R = sum K |x - z|^2 and K relative to the
    offset of the SEGMENT's own minimum; merged online, turned into the column's moment terms with x_n as the origin, mapped back
    to the source's frame - against the reference's row-side moments and the oracle's M-step."""
    rng = np.random.default_rng(11)
    m, n = 600, 800
    y = rng.normal(size=(m, 3))
    rot, t = _rot(-0.2, 0.4, 0.1), np.array([-0.2, 0.15, 0.3])
    x = (scale * 0.99) * y[rng.integers(0, m, n)] @ _rot(-0.22, 0.37, 0.12).T + t + 0.04 * rng.normal(size=(n, 3))
    z = scale * y @ rot.T + t
    es = co.expectation_step(z, x, sigma2, w)
    pt1, p1, px, n_p = es
    ref = dict(S0=n_p, Sx=px.sum(0), Sy=y.T @ p1, Sxy=px.T @ y, trSyy=float(np.sum(p1 * np.sum(y * y, axis=1))),
               Sxx=float(np.sum(pt1 * np.sum(x * x, axis=1))))
    kk = -np.log2(np.e) / (2.0 * sigma2)
    c = (2.0 * np.pi * sigma2) ** 1.5 * w / (1.0 - w) * m / n if w > 0 else 0.0
    bounds = np.linspace(0, m, nseg + 1).astype(int)
    mom = np.zeros(24)
    for j in range(n):
        gmin, goff = np.inf, np.inf
        A, U, R = 0.0, np.zeros(3), 0.0
        for s in range(nseg):
            zs = z[bounds[s]:bounds[s + 1]]
            d = x[j] - zs
            d2 = np.sum(d * d, axis=1)
            pm = d2.min()
            off = float(_col_offset(kk, pm))
            K = np.exp2(kk * d2 + off)
            a, u, r = K.sum(), K @ d, K @ d2
            if pm < gmin:                                      # k_colfinal_resid's merge
                noff = float(_col_offset(kk, pm))
                f = 0.0 if np.isinf(goff) else np.exp2(noff - goff)
                A, U, R = A * f, U * f, R * f
                gmin, goff = pm, noff
            f = np.exp2(goff - off)
            A, U, R = A + a * f, U + u * f, R + r * f
        den = A * np.exp2(-goff)
        pd = den / (den + c)
        qn = pd / A
        pz = pd * x[j] - qn * U
        mom[0] += pd
        mom[1:4] += pd * x[j]
        mom[4:7] += pz
        mom[7:16] += np.outer(x[j], pz).ravel()
        mom[16] += pd * (x[j] @ x[j]) + qn * (R - 2.0 * x[j] @ U)
        mom[22] += pd * (x[j] @ x[j])
    S0, Sx, Sz, Sxz = mom[0], mom[1:4], mom[4:7], mom[7:16].reshape(3, 3)
    Sy = rot.T @ (Sz - S0 * t) / scale
    Sxy = (Sxz - np.outer(Sx, t)) @ rot / scale
    trSyy = (mom[16] - 2.0 * t @ Sz + S0 * (t @ t)) / scale ** 2
    assert abs(S0 - ref["S0"]) < 1e-11 * ref["S0"]
    assert np.max(np.abs(Sx - ref["Sx"])) < 1e-10 * ref["S0"]
    assert np.max(np.abs(Sy - ref["Sy"])) < 1e-10 * ref["S0"]
    assert np.max(np.abs(Sxy - ref["Sxy"])) < 1e-10 * ref["S0"]
    assert abs(trSyy - ref["trSyy"]) < 1e-10 * ref["S0"]
    assert abs(mom[22] - ref["Sxx"]) < 1e-10 * ref["S0"]
    mu_x, mu_y = Sx / S0, Sy / S0
    a = Sxy - np.outer(Sx, mu_y)
    u, _, vh = np.linalg.svd(a)
    cdiag = np.ones(3)
    cdiag[-1] = np.linalg.det(u @ vh)
    r_new = (u * cdiag) @ vh
    tr_atr = np.trace(a.T @ r_new)
    s_new = tr_atr / (trSyy - S0 * (mu_y @ mu_y))
    sigma2_new = (mom[22] - S0 * (mu_x @ mu_x) - s_new * tr_atr) / (S0 * 3)
    p, s2, _q = co.mstep_rigid(y, x, es)
    assert np.max(np.abs(r_new - p["rot"])) < 1e-9 and abs(s_new - p["scale"]) < 1e-9
    assert abs(sigma2_new - s2) < 1e-8 * s2
