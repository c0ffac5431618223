"""CPU model of the non-rigid path's kernel factor - what the GPU tests of test_nonrigid_lowrank_gpu.py rest on.

The product keeps G = F F^T (pivoted Cholesky, columns evaluated in fp64) instead of the reference's float32 M x M matrix
(transformation.py:91-99) and solves the M-step (cpd.py:284-303) through an r x r system.  Shown here with numpy, against the
oracle's own M-step:
  * the factor reproduces the exact kernel matrix to the tolerance of the factorisation, at a rank far below M;
  * the push-through solve  W = (B - D F z) / c,  (c I + F^T D F) z = F^T B  equals numpy's solve on the exact matrix, and
    F^T W = z (the identity that makes G W = F z free);
  * replacing the reference's float32 G by the exact one moves a whole registration by less than the non-rigid tolerances
    (the float32 rounding of G is the ONLY difference between the two paths)."""
import numpy as np

from oracle import cpd_numpy as co


def pivoted_cholesky(y, beta, tol=1e-14, max_rank=2048):
    """The algorithm of k_pchol_step (csrc/cpd_nonrigid.hip), column by column."""
    m = len(y)
    d = np.ones(m)
    f = np.zeros((m, min(max_rank, m)))
    j = 0
    while j < f.shape[1]:
        p = int(np.argmax(d))
        if d[p] <= tol:
            break
        g = np.exp(-np.sum((y - y[p]) ** 2, axis=1) / (2.0 * beta))
        col = (g - f[:, :j] @ f[p, :j]) / np.sqrt(d[p])
        col[p] = np.sqrt(d[p])
        f[:, j] = col
        d = np.maximum(d - col * col, 0.0)
        d[p] = 0.0
        j += 1
    return f[:, :j], float(d.max())


def _clouds(m, seed):
    from probreg_amd import synthetic

    src, tgt = synthetic.nonrigid_pair(m + 100, m=m, seed=seed)
    return src, tgt


def test_factor_is_exact_at_low_rank():
    src, _ = _clouds(1500, 3)
    g = np.exp(-co._sqdist(src, src) / (2.0 * 2.0))
    f, rest = pivoted_cholesky(src, 2.0)
    assert f.shape[1] < 260 and rest <= 1e-14          # C3-style cloud, beta = 2: rank ~170, whatever M is
    assert np.max(np.abs(f @ f.T - g)) < 5e-14
    f_narrow, _ = pivoted_cholesky(src, 0.05)
    assert f_narrow.shape[1] > 3 * f.shape[1]          # the rank is a property of the kernel width, not of M


def test_push_through_solve_equals_dense_solve():
    src, tgt = _clouds(1200, 4)
    f, _ = pivoted_cholesky(src, 2.0)
    g = np.exp(-co._sqdist(src, src) / (2.0 * 2.0))
    sigma2 = co.squared_kernel_sum(src, tgt) * 0.05
    pt1, p1, px, n_p = co.expectation_step(src, tgt, sigma2, 0.0)
    lmd = 2.0
    c = lmd * sigma2
    b = px - p1[:, None] * src
    z = np.linalg.solve(c * np.identity(f.shape[1]) + f.T @ (p1[:, None] * f), f.T @ b)
    w = (b - p1[:, None] * (f @ z)) / c
    want = np.linalg.solve(p1[:, None] * g + c * np.identity(len(src)), b)
    assert np.max(np.abs(w - want)) < 1e-8 * np.max(np.abs(want))
    assert np.max(np.abs(f.T @ w - z)) < 1e-9 * np.max(np.abs(z))   # => G W = F z, no second pass over F


def test_exact_kernel_moves_the_registration_less_than_the_tolerances():
    src, tgt = _clouds(1000, 5)
    g32 = co.rbf_kernel(src, src, 2.0)                  # the reference's float32 matrix
    gex = np.exp(-co._sqdist(src, src) / (2.0 * 2.0))   # what the factor reproduces
    assert 1e-8 < np.max(np.abs(g32 - gex)) < 3e-7      # float32 rounding of entries in (0, 1]
    out = {}
    for name, g in (("f32", g32), ("exact", gex)):
        sigma2 = co.squared_kernel_sum(src, tgt)
        params = dict(w=np.zeros_like(src))
        for _ in range(12):
            es = co.expectation_step(src + g @ params["w"], tgt, sigma2, 0.0)
            params, sigma2, _ = co.mstep_nonrigid(src, tgt, es, sigma2, g, 2.0)
        out[name] = (sigma2, src + g @ params["w"])
    ext = np.max(np.abs(out["f32"][1] - out["f32"][1].mean(0)))
    assert np.max(np.abs(out["exact"][1] - out["f32"][1])) < 2e-5 * ext     # T(Y): tolerance 1e-4
    assert abs(out["exact"][0] - out["f32"][0]) < 5e-6 * out["f32"][0]      # sigma2: tolerance 1e-5
