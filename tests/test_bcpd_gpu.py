"""BCPD on the GPU against the reference's golden outputs and the numpy oracle.

Tolerances: the E-step arithmetic is float32 on the GPU (float64 in the reference) -> 2e-5 of the largest entry;
transformations 1e-4, as for CPD.  The M-step fixtures use well-separated points because the reference inverts the
float32 kernel matrix in float32 (bcpd.py:108) while the GPU never forms G^-1 (see tests/test_oracle_bcpd.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, Golden, rel_err

pytestmark = pytest.mark.gpu

TOL_E = 2e-5
TOL_T = 1e-4

ESTEP_CASES = ["uniform_alpha_w0", "alpha_vec_w0.1", "small_sigma2_w0.3", "planar_w0.05"]
REG_CASES = ["grid48_default", "grid48_w0.1_k5", "grid48_lmd20_k1", "grid120_w0.05_k6"]


@pytest.fixture(scope="module")
def bcpd_golden():
    return Golden(os.path.join(GOLDEN_DIR, "bcpd_golden.npz"))


def reg_kwargs(c):
    kw = {}
    for k in ("w", "maxiter", "tol", "lmd", "k", "gamma"):
        if "arg_" + k in c:
            kw[k] = int(c["arg_" + k]) if k == "maxiter" else float(c["arg_" + k])
    return kw


@pytest.mark.parametrize("name", ESTEP_CASES)
def test_estep_matches_reference(bcpd_golden, name):
    from probreg_amd import bcpd

    c = bcpd_golden.case("estep/" + name)
    reg = bcpd.CombinedBCPD(c["t_source"])
    es = reg.expectation_step(c["t_source"], c["target"], c["scale"], c["alpha"], c["sigma_diag"], c["sigma2"], c["w"])
    assert rel_err(es.nu_d, c["out_nu_d"]) < TOL_E
    assert rel_err(es.nu, c["out_nu"]) < TOL_E
    assert rel_err(es.px, c["out_px"]) < TOL_E
    assert abs(es.n_p - np.sum(c["out_nu"])) < TOL_E * np.sum(c["out_nu"])
    ok = c["out_nu"] > 1e-3 * np.max(c["out_nu"])
    assert rel_err(es.x_hat[ok], c["out_x_hat"][ok]) < 5e-5
    # a full sigma_mat is accepted too and only its diagonal matters (bcpd.py:61)
    es2 = reg.expectation_step(c["t_source"], c["target"], c["scale"], c["alpha"], np.diag(c["sigma_diag"]), c["sigma2"],
                               c["w"])
    assert np.array_equal(es2.nu, es.nu)


def test_estep_matches_oracle_at_size():
    from oracle import bcpd_numpy as bo
    from probreg_amd import bcpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(5000, m=3500, seed=41)
    rng = np.random.default_rng(2)
    alpha = rng.dirichlet(np.full(src.shape[0], 1.5))
    sd = rng.uniform(1e-5, 5e-3, src.shape[0])
    reg = bcpd.CombinedBCPD(src)
    for sigma2, w in ((0.05, 0.0), (2e-3, 0.1), (2e-4, 0.2)):  # the last one is deep in the culled regime
        es = reg.expectation_step(src, tgt, 1.05, alpha, sd, sigma2, w)
        ref = bo.expectation_step(src, tgt, 1.05, alpha, sd, sigma2, w)
        assert rel_err(es.nu_d, ref.nu_d) < TOL_E, sigma2
        assert rel_err(es.nu, ref.nu) < TOL_E, sigma2
        assert rel_err(es.px, ref.px) < TOL_E, sigma2


def test_weights_must_be_normalised():
    from probreg_amd import engine

    plan = engine.CpdPlan()
    try:
        plan.set_source(np.random.default_rng(0).normal(size=(100, 3)))
        with pytest.raises(ValueError):
            plan.set_source_weights(np.full(100, 0.5))
        plan.set_source_weights(np.full(100, -0.5))
        plan.set_source_weights(None)
    finally:
        plan.close()


def test_mstep_matches_reference(bcpd_golden):
    from probreg_amd import bcpd, transformation as tf

    c = bcpd_golden.case("mstep/grid120")
    reg = bcpd.CombinedBCPD(c["source"])
    es = bcpd.EstepResult(c["nu_d"], c["nu"], float(np.sum(c["nu"])), c["px"], c["x_hat"])
    ms = reg.maximization_step(c["target"], tf.RigidTransformation(np.identity(3), np.zeros(3), 1.0), es, c["sigma2_p"])
    rt = ms.transformation.rigid_trans
    assert rel_err(rt.rot, c["out_rot"]) < 1e-6
    assert rel_err(rt.t, c["out_t"]) < 1e-5
    assert abs(rt.scale - c["out_scale"]) < 1e-6
    assert rel_err(ms.transformation.v, c["out_v"]) < 1e-4
    assert rel_err(ms.sigma_mat, c["out_sigma_diag"]) < 1e-5
    assert rel_err(ms.alpha, c["out_alpha"]) < 1e-12
    assert abs(ms.sigma2 - c["out_sigma2"]) < 1e-6 * c["out_sigma2"]


def test_solve_matches_dense_inverse():
    """prg_cpd_bcpd_solve against the explicit float64 inverse, including points without any support (nu = 0)
    and a size that is not a multiple of the 128 / 512 blocking."""
    from oracle import bcpd_numpy as bo
    from probreg_amd import engine

    rng = np.random.default_rng(7)
    m = 777
    src = rng.uniform(-12.0, 12.0, (m, 3))
    nu = rng.uniform(0.0, 2.0, m)
    nu[rng.choice(m, 60, replace=False)] = 0.0
    resid = rng.normal(0.0, 0.3, (m, 3))
    lmd, cfac = 2.0, 37.5
    plan = engine.CpdPlan()
    try:
        plan.set_source(src)
        plan.bcpd_build_g(1.0)
        v, sd = plan.bcpd_solve(lmd, cfac, resid, nu)
    finally:
        plan.close()
    g = bo.inverse_multiquadric_kernel(src, src).astype(np.float64)
    sigma = np.linalg.inv(lmd * np.linalg.inv(g) + cfac * np.diag(nu))
    assert rel_err(sd, np.diag(sigma)) < 1e-6
    assert rel_err(v, cfac * sigma @ (nu[:, None] * resid)) < 1e-6


@pytest.mark.parametrize("name", REG_CASES)
def test_registration_matches_reference(bcpd_golden, name):
    from probreg_amd import bcpd

    c = bcpd_golden.case("reg/" + name)
    hist = []
    trans = bcpd.registration_bcpd(c["source"], c["target"], callbacks=[hist.append], **reg_kwargs(c))
    assert len(hist) == c["out_niter"]
    assert rel_err(trans.rigid_trans.rot, c["out_rot"]) < TOL_T
    assert abs(trans.rigid_trans.scale - c["out_scale"]) < TOL_T
    assert rel_err(trans.transform(c["source"]), c["out_tsource"]) < TOL_T


def test_registration_matches_oracle_at_size():
    from oracle import bcpd_numpy as bo
    from probreg_amd import bcpd

    rng = np.random.default_rng(11)
    g = np.stack(np.meshgrid(np.arange(10), np.arange(9), np.arange(8), indexing="ij"), axis=-1).reshape(-1, 3)
    src = g * 3.0 + rng.uniform(-0.7, 0.7, g.shape)
    src -= src.mean(axis=0)
    a = np.deg2rad(8.0)
    r = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    tgt = 0.97 * (src + 0.4 * np.sin(0.3 * src[:, [2, 0, 1]])) @ r.T + np.array([0.5, 0.2, -0.4])
    tgt = tgt + rng.normal(0.0, 0.05, tgt.shape)
    tgt = np.concatenate([tgt, rng.uniform(tgt.min(axis=0), tgt.max(axis=0), (80, 3))], axis=0)
    trans = bcpd.registration_bcpd(src, tgt, w=0.1, maxiter=8, tol=-1.0)
    res, niter = bo.registration(src, tgt, w=0.1, maxiter=8, tol=-1.0, inv_dtype=np.float64)
    ts = res.scale * np.dot(src + res.v, res.rot.T) + res.t
    assert rel_err(trans.transform(src), ts) < TOL_T
    assert abs(trans.rigid_trans.scale - res.scale) < TOL_T
    assert rel_err(trans.rigid_trans.rot, res.rot) < TOL_T


def test_math_utils_helpers():
    from scipy.spatial import cKDTree

    from oracle import bcpd_numpy as bo
    from probreg_amd import math_utils as mu, synthetic

    src, tgt, _ = synthetic.rigid_pair(3000, m=2100, seed=5)
    k = mu.inverse_multiquadric_kernel(src[:300], tgt[:200], 0.7)
    assert k.dtype == np.float32 and k.shape == (300, 200)
    assert rel_err(k, bo.inverse_multiquadric_kernel(src[:300], tgt[:200], 0.7)) < 3e-7
    want = np.sum(cKDTree(tgt).query(src)[0]) / src.shape[0]
    assert abs(mu.compute_rmse(src, tgt) - want) < 1e-6 * want
    assert abs(mu.compute_rmse(src + 1000.0, cKDTree(tgt + 1000.0)) - want) < 1e-5 * want  # tree accepted, offsets centred
    assert abs(mu.compute_rmse(src[:, :2], tgt[:, :2]) - np.mean(cKDTree(tgt[:, :2]).query(src[:, :2])[0])) < 1e-6
