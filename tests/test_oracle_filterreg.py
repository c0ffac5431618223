"""FilterReg oracle (C lattice restatement + numpy driver) against fixtures produced by the reference's
own filterreg.py + vendored permutohedral.cpp (tests/golden/make_golden.py filterreg).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, Golden
from oracle import filterreg_numpy as fo
from oracle import permutohedral as ph


@pytest.fixture(scope="module")
def fr_golden():
    return Golden(os.path.join(GOLDEN_DIR, "filterreg_golden.npz"))


def test_lattice_restatement_is_bit_exact(fr_golden):
    for name in fr_golden.group("lattice"):
        c = fr_golden.case("lattice/" + name)
        lat = ph.Lattice(c["points"], "blur1" in name, prefer_ref=False)
        assert not lat.is_ref
        assert lat.lattice_size == c["size"], name
        for ch in (1, 3, 5):
            got = lat.filter(c["values_ch%d" % ch])
            assert np.array_equal(got, c["out_ch%d" % ch]), (name, ch)


def test_vendored_reference_build_agrees_when_present(fr_golden):
    if not ph.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    c = fr_golden.case("lattice/d3_blur1")
    lat = ph.Lattice(c["points"], True, prefer_ref=True)
    assert lat.is_ref and lat.lattice_size == c["size"]
    assert np.array_equal(lat.filter(c["values_ch3"]), c["out_ch3"])


def test_reference_unit_test_gaussian_filtering():
    """Port of the reference's tests/test_gaussian_filtering.py:7-18: the 1-D lattice ratio out0/out1
    approximates the direct Gauss transform ratio with h = sqrt(2) to rtol 0.3."""
    rng = np.random.default_rng(3)
    pts = rng.uniform(0.0, 10.0, (100, 1))
    v0 = np.ones((100, 1))
    v1 = rng.uniform(0.0, 1.0, (100, 1))
    lat = ph.Lattice(pts, True)
    out0, out1 = lat.filter(v0), lat.filter(v1)
    h2 = 2.0
    k = np.exp(-((pts - pts.T) ** 2) / h2)
    want = (k @ v1) / (k @ v0)
    assert np.allclose(out1 / out0, want, rtol=0.3)


REG = ["bunny_default", "bunny_update_sigma2", "bunny_update_sigma2_w005_k8", "bunny_fixed_sigma2_k5",
       "synth_5k_outliers_k6", "synth_ragged_k4", "fish2d_k10"]


@pytest.mark.parametrize("name", REG)
def test_registration_matches_reference(fr_golden, name):
    c = fr_golden.case("reg/" + name)
    kw = {}
    for k in ("sigma2", "update_sigma2", "w", "maxiter", "tol"):
        if "arg_" + k in c:
            kw[k] = c["arg_" + k]
    if "maxiter" in kw:
        kw["maxiter"] = int(kw["maxiter"])
    if "update_sigma2" in kw:
        kw["update_sigma2"] = bool(kw["update_sigma2"])
    rot, t, s2, q, niter = fo.registration(c["source"], c["target"], **kw)
    assert niter == c["out_niter"]
    # exact up to the float32 summation order inside the sigma2 initialiser (one float32 ulp when the source
    # has more rows than the oracle's block size), which then propagates at the 1e-7 level
    assert np.max(np.abs(rot - c["out_rot"])) < 1e-6
    assert np.max(np.abs(t - c["out_t"])) < 1e-6
    assert abs(s2 - c["out_sigma2"]) <= 1e-6 * abs(c["out_sigma2"])
    assert abs(q - c["out_q"]) <= 1e-6 * abs(c["out_q"])


def test_kabsch_restatement_properties():
    """cc/kabsch.cc: weights enter the centroids linearly and the covariance squared; recovers an exact motion."""
    rng = np.random.default_rng(5)
    a = rng.normal(size=(400, 3))
    th = 0.4
    r = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    b = a @ r.T + np.array([0.3, -0.2, 0.1])
    w = rng.uniform(0.5, 1.5, 400)
    rr, tt = fo.kabsch_f32(a, b, w)
    assert rr.dtype == np.float32 and np.allclose(rr, r, atol=2e-6) and np.allclose(tt, [0.3, -0.2, 0.1], atol=2e-6)
    r2, t2 = fo.kabsch2d_f32(a[:, :2], b[:, :2], w)
    assert np.allclose(r2, r[:2, :2], atol=2e-6)
    r0, t0 = fo.kabsch_f32(a, b, np.zeros(400))
    assert np.array_equal(r0, np.identity(3)) and np.array_equal(t0, np.zeros(3))


def test_pt2pl_registration_matches_reference(fr_golden):
    """Point-to-plane objective: filterreg.py:183-186 over the restated cc/point_to_plane.cc + se3_op.twist_mul."""
    for name in fr_golden.group("pt2pl"):
        c = fr_golden.case("pt2pl/" + name)
        kw = {k[4:]: c[k] for k in c if k.startswith("arg_")}
        if "maxiter" in kw:
            kw["maxiter"] = int(kw["maxiter"])
        if "update_sigma2" in kw:
            kw["update_sigma2"] = bool(kw["update_sigma2"])
        rot, t, s2, q, _ = fo.registration(c["source"], c["target"], target_normals=c["normals"], objective_type="pt2pl", **kw)
        assert np.max(np.abs(rot - c["out_rot"])) < 1e-6 and np.max(np.abs(t - c["out_t"])) < 1e-6, name
        assert abs(s2 - c["out_sigma2"]) <= 1e-6 * abs(c["out_sigma2"]), name
        assert abs(q - c["out_q"]) <= 1e-6 * abs(c["out_q"]), name


@pytest.mark.parametrize("d", [5, 33])
@pytest.mark.parametrize("blur", [1, 0])
def test_feature_lattice_restatement_is_bit_exact(feature_golden, d, blur):
    """The plain-C lattice at feature dimensions (FPFH is d = 33, features.py:28-51) against the vendored
    permutohedral.cpp's own outputs: same vertex count, bit-identical filter results."""
    from oracle import permutohedral as ph

    c = feature_golden.case("lattice/d%d_blur%d" % (d, blur))
    lat = ph.Lattice(c["points"], bool(blur))
    assert lat.lattice_size == int(c["size"])
    for ch in (1, 3):
        assert np.array_equal(lat.filter(c["values_ch%d" % ch]), c["out_ch%d" % ch])


@pytest.mark.parametrize("name", ["feat8_update_k5", "feat33_fixed_k4", "feat8_auto_sigma2_k3"])
def test_feature_registration_matches_reference(feature_golden, name):
    """oracle.filterreg_numpy.registration with a non-identity feature_fn against the reference's
    registration_filterreg(feature_fn=...) (filterreg.py:121, 125-133)."""
    from conftest import golden_feature_map
    from oracle import filterreg_numpy as fo

    c = feature_golden.case("reg/" + name)
    kw = {k[4:]: c[k] for k in c if k.startswith("arg_")}
    if "maxiter" in kw:
        kw["maxiter"] = int(kw["maxiter"])
    if "update_sigma2" in kw:
        kw["update_sigma2"] = bool(kw["update_sigma2"])
    rot, t, s2, q, _ = fo.registration(c["source"], c["target"], feature_fn=golden_feature_map(c), **kw)
    assert np.max(np.abs(rot - c["out_rot"])) < 2e-6 and np.max(np.abs(t - c["out_t"])) < 2e-6
    assert abs(s2 - c["out_sigma2"]) <= 2e-6 * c["out_sigma2"]
    assert abs(q - c["out_q"]) <= 1e-5 * abs(c["out_q"])
