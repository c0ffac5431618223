"""Edge cases of the GPU paths: plan reuse, extreme aspect ratios, alignment boundaries of the kernels' tiling,
degenerate inputs, heavy outliers, deep late-regime (culled) iterations.  Everything is checked against the numpy
oracle on the same inputs; tolerances as in test_cpd_gpu.py (transform 1e-4, sigma2 1e-5)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_TF = 1e-4
TOL_SIGMA2 = 1e-5


def _check(kind, res, p, s2):
    tr = res.transformation
    lin = tr.rot if kind == "rigid" else tr.b
    assert rel_err(lin, p["rot"] if kind == "rigid" else p["b"]) < TOL_TF, kind
    assert np.max(np.abs(tr.t - p["t"])) < TOL_TF * max(1.0, np.max(np.abs(p["t"]))), kind
    if kind == "rigid":
        assert abs(tr.scale - p["scale"]) < TOL_TF * p["scale"]
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2, (kind, res.sigma2, s2)


@pytest.mark.parametrize("m,n", [(127, 129), (128, 128), (513, 255), (1025, 2047), (2049, 511)])
def test_registration_across_tile_boundaries(m, n):
    """Cloud sizes just below / at / above the 128-point wave tiles, 256-point segment quantum and 512-point segments."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=m + n)
    for kind in ("rigid", "affine"):
        res = cpd.registration_cpd(src, tgt, kind, w=0.1, maxiter=12, tol=-1.0)
        p, s2, q, _ = co.registration(kind, src, tgt, w=0.1, maxiter=12, tol=-1.0, closed_form_init=True)
        _check(kind, res, p, s2)


@pytest.mark.parametrize("m,n", [(3, 6000), (6000, 4), (40, 9000), (9000, 60)])
def test_extreme_aspect_ratios(m, n):
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=7)
    got = cpd.RigidCPD().expectation_step(src, tgt, 0.05, 0.15)
    want = co.expectation_step(src, tgt, 0.05, 0.15)
    assert np.max(np.abs(got.pt1 - want.pt1)) < 2e-6
    assert rel_err(got.p1, want.p1) < 1e-5 and rel_err(got.px, want.px) < 1e-5
    if min(m, n) >= 40:
        res = cpd.registration_cpd(src, tgt, "rigid", w=0.05, maxiter=10, tol=-1.0)
        p, s2, q, _ = co.registration("rigid", src, tgt, w=0.05, maxiter=10, tol=-1.0, closed_form_init=True)
        _check("rigid", res, p, s2)


def test_deep_late_regime_matches_oracle():
    """40 iterations: sigma2 falls by four orders of magnitude, the sweeps end up skipping almost every block.
    Segment / plane layout of this size differs from C1's (11 column segments of 256 points, 3 planes)."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(3500, m=3000, seed=91)
    for kind, w in (("rigid", 0.0), ("affine", 0.1)):
        res = cpd.registration_cpd(src, tgt, kind, w=w, maxiter=40, tol=-1.0)
        p, s2, q, _ = co.registration(kind, src, tgt, w=w, maxiter=40, tol=-1.0, closed_form_init=True, c_estep=True)
        _check(kind, res, p, s2)
    assert res.sigma2 < 1e-3


def test_heavy_outliers_and_large_w():
    """A third of the target is uniform clutter far from the object and w = 0.9: most columns end up dominated by
    the uniform term, some die completely (den underflows) in late iterations."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(2000, m=1500, seed=12)
    rng = np.random.default_rng(3)
    clutter = rng.uniform(-6.0, 6.0, (1000, 3))
    tgt = np.concatenate([tgt, clutter], axis=0)
    rng.shuffle(tgt, axis=0)
    res = cpd.registration_cpd(src, tgt, "rigid", w=0.9, maxiter=25, tol=-1.0)
    p, s2, q, _ = co.registration("rigid", src, tgt, w=0.9, maxiter=25, tol=-1.0, closed_form_init=True)
    _check("rigid", res, p, s2)


def test_identical_clouds():
    """source == target: the transform stays the identity while sigma2 collapses to the float32-eps clamp."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, _, _ = synthetic.rigid_pair(10, m=900, seed=5)
    res = cpd.registration_cpd(src, src.copy(), "rigid", maxiter=30, tol=-1.0)
    p, s2, q, _ = co.registration("rigid", src, src.copy(), maxiter=30, tol=-1.0, closed_form_init=True)
    assert rel_err(res.transformation.rot, np.identity(3)) < 1e-5
    assert np.max(np.abs(res.transformation.t)) < 1e-5
    assert abs(res.transformation.scale - 1.0) < 1e-5
    assert res.sigma2 <= max(2.0 * s2, 2.0 * np.finfo(np.float32).eps)


def test_registrar_reuse_with_new_targets_and_sources():
    """One registrar object, several registrations: bigger target (buffers grow), smaller target, new source."""
    from probreg_amd import cpd, synthetic

    src, tgt_a, _ = synthetic.rigid_pair(1200, m=1000, seed=21)
    _, tgt_b, _ = synthetic.rigid_pair(5000, m=1000, seed=22)
    src2, tgt_c, _ = synthetic.rigid_pair(800, m=2300, seed=23)
    reg = cpd.RigidCPD(src)
    runs = [(src, tgt_a), (src, tgt_b), (src, tgt_a), (src2, tgt_c), (src, tgt_b)]
    for s, t in runs:
        if s is not reg._source and not np.array_equal(s, reg._source):
            reg.set_source(s)
        got = reg.registration(t, w=0.05, maxiter=10, tol=-1.0)
        fresh = cpd.RigidCPD(s).registration(t, w=0.05, maxiter=10, tol=-1.0)
        assert np.array_equal(got.transformation.rot, fresh.transformation.rot)
        assert got.sigma2 == fresh.sigma2


def test_bcpd_two_dimensional_and_w0():
    from oracle import bcpd_numpy as bo
    from probreg_amd import bcpd

    rng = np.random.default_rng(8)
    g = np.stack(np.meshgrid(np.arange(9), np.arange(8), indexing="ij"), axis=-1).reshape(-1, 2)
    src = g * 3.0 + rng.uniform(-0.6, 0.6, g.shape)
    src -= src.mean(axis=0)
    th = 0.15
    r = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    tgt = 1.03 * (src + 0.3 * np.sin(0.4 * src[:, ::-1])) @ r.T + np.array([0.4, -0.3]) + rng.normal(0.0, 0.04, src.shape)
    for w in (0.0, 0.1):
        trans = bcpd.registration_bcpd(src, tgt, w=w, maxiter=6, tol=-1.0)
        res, _ = bo.registration(src, tgt, w=w, maxiter=6, tol=-1.0, inv_dtype=np.float64)
        ts = res.scale * np.dot(src + res.v, res.rot.T) + res.t
        assert trans.rigid_trans.rot.shape == (2, 2)
        assert rel_err(trans.transform(src), ts) < TOL_TF
        assert abs(trans.rigid_trans.scale - res.scale) < TOL_TF


@pytest.mark.parametrize("workload", ["C1_rigid_100k", "C2_affine_200k"])
def test_full_size_identities_through_every_regime(workload):
    """BASELINE configs at full size, 30 EM iterations (dense -> mid -> late: the culled sweeps end up skipping 97 %
    of the blocks).  With w = 0 every column of P sums to one, so after EVERY E-step n_p = N, sum_m px_m = sum_n x_n
    and sum_n pt1_n |x_n|^2 = sum_n |x_n|^2 - size-independent checks of the whole E-step; the M-step must keep the
    rotation orthonormal (rigid) and sigma2 must fall monotonically on this data."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    if workload.startswith("C1"):
        n, kind = 100000, _lib.PRG_TF_RIGID
        src, tgt, _ = synthetic.rigid_pair(n, seed=0)
    else:
        n, kind = 200000, _lib.PRG_TF_AFFINE
        src, tgt, _ = synthetic.affine_pair(n, seed=0)
    s32, t32 = (src - src.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
    t64 = t32.astype(np.float64)
    sx, sxx = t64.sum(0), np.sum(t64 * t64)
    plan = CpdPlan()
    plan.set_source(s32)
    plan.set_target(t32)
    plan.init_sums()
    plan.init_params(None)
    prev = np.inf
    for it in range(30):
        plan.estep(0.0)
        mom = plan.get_moments()
        assert abs(mom[0] - n) < 3e-6 * n, (it, mom[0])
        assert np.max(np.abs(mom[1:4] - sx)) < 3e-6 * n, it
        assert abs(mom[22] - sxx) < 3e-6 * sxx, it
        plan.mstep(kind, True)
        p = plan.get_params()
        assert p[13] < prev
        prev = p[13]
        if kind == _lib.PRG_TF_RIGID:
            rot = p[:9].reshape(3, 3)
            assert np.allclose(rot @ rot.T, np.eye(3), atol=1e-12)
    assert prev < 2e-4
    plan.close()


def test_million_point_clouds():
    """N = M = 1e6 (1e12 pairs per E-step; the reference's M x N float64 matrix would be 8 TB): the column-sum
    identities hold and the E-step costs what 100x C1 predicts (linear in M x N)."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    n = 1000000
    src, tgt, _ = synthetic.rigid_pair(n, seed=0)
    s32, t32 = (src - src.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
    t64 = t32.astype(np.float64)
    sx, sxx = t64.sum(0), np.sum(t64 * t64)
    plan = CpdPlan()
    plan.set_source(s32)
    plan.set_target(t32)
    plan.init_sums()
    plan.init_params(None)
    for it in range(2):
        ms = plan.estep_timed(0.0)
        mom = plan.get_moments()
        assert abs(mom[0] - n) < 3e-6 * n
        assert np.max(np.abs(mom[1:4] - sx)) < 3e-6 * n
        assert abs(mom[22] - sxx) < 3e-6 * sxx
        assert ms["total"] < 1000.0  # ~400 ms on an MI355X
        plan.mstep(_lib.PRG_TF_RIGID, True)
    plan.close()


def test_singular_affine_raises_like_the_reference():
    """Two source points cannot determine a 2-D affine map: np.linalg.solve raises in the reference (cpd.py:237),
    the device solve leaves non-finite numbers, and the wrapper turns those into the same exception."""
    from probreg_amd import cpd

    rng = np.random.default_rng(0)
    src = rng.normal(size=(2, 2))
    tgt = rng.normal(size=(300, 2))
    with pytest.raises(np.linalg.LinAlgError):
        cpd.registration_cpd(src, tgt, "affine", maxiter=5, tol=-1.0)
