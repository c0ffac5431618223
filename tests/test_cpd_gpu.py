"""Parity of the HIP CPD path (through the C ABI) with the reference: golden fixtures produced by
the reference's own code, the numpy oracle on seeded inputs, and size-independent identities at
the BASELINE.json sizes.  Tolerances are the north-star's: transform within 1e-4 relative,
sigma2 within 1e-5 relative."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_TF = 1e-4
TOL_SIGMA2 = 1e-5


def _kind(name):
    return "nonrigid" if "nonrigid" in name else ("affine" if "affine" in name else "rigid")


def _kwargs(c):
    kw = {}
    for k in ("w", "maxiter", "tol", "update_scale"):
        if "arg_" + k in c:
            kw[k] = c["arg_" + k]
    if "maxiter" in kw:
        kw["maxiter"] = int(kw["maxiter"])
    if "update_scale" in kw:
        kw["update_scale"] = bool(kw["update_scale"])
    return kw


def _check_rigid(res, rot, t, scale, sigma2):
    tr = res.transformation
    assert rel_err(tr.rot, rot) < TOL_TF
    assert np.max(np.abs(tr.t - t)) < TOL_TF * max(1.0, np.max(np.abs(t)))
    assert abs(tr.scale - scale) < TOL_TF * abs(scale)
    assert abs(res.sigma2 - sigma2) <= TOL_SIGMA2 * abs(sigma2)


def _check_affine(res, b, t, sigma2):
    tr = res.transformation
    assert rel_err(tr.b, b) < TOL_TF
    assert np.max(np.abs(tr.t - t)) < TOL_TF * max(1.0, np.max(np.abs(t)))
    assert abs(res.sigma2 - sigma2) <= TOL_SIGMA2 * abs(sigma2)


FIXED_ITER_CASES = [
    "bunny_rigid_noscale_w01_k10", "synth_rigid_2k_k1", "synth_rigid_2k_k3", "synth_rigid_2k_k10",
    "synth_rigid_2k_w02_k5", "synth_affine_2k_k1", "synth_affine_2k_k10", "synth_rigid_ragged_k6",
]


@pytest.mark.parametrize("name", FIXED_ITER_CASES)
def test_registration_fixed_iterations_vs_reference(cpd_golden, name):
    from probreg_amd import cpd

    c = cpd_golden.case("reg/" + name)
    kind = _kind(name)
    res = cpd.registration_cpd(c["source"], c["target"], kind, **_kwargs(c))
    if kind == "rigid":
        _check_rigid(res, c["out_rot"], c["out_t"], c["out_scale"], c["out_sigma2"])
    else:
        _check_affine(res, c["out_b"], c["out_t"], c["out_sigma2"])
    assert abs(res.q - c["out_q"]) <= 1e-4 * abs(c["out_q"]) + 1e-2


@pytest.mark.parametrize("name", ["bunny_rigid_default", "bunny_affine_default", "fish_rigid_default",
                                  "fish_affine_default"])
def test_registration_defaults_vs_reference(cpd_golden, name):
    """Default arguments (tol=1e-3 on an absolute q): iteration counts may differ by one or two
    between an fp32 and an fp64 E-step, the converged answer may not."""
    from probreg_amd import cpd

    c = cpd_golden.case("reg/" + name)
    kind = _kind(name)
    niter = [0]
    res = cpd.registration_cpd(c["source"], c["target"], kind,
                               callbacks=[lambda t: niter.__setitem__(0, niter[0] + 1)])
    assert abs(niter[0] - c["out_niter"]) <= 2
    if kind == "rigid":
        _check_rigid(res, c["out_rot"], c["out_t"], c["out_scale"], c["out_sigma2"])
    else:
        _check_affine(res, c["out_b"], c["out_t"], c["out_sigma2"])


def test_estep_vs_reference(cpd_golden):
    from probreg_amd import cpd

    obj = cpd.RigidCPD()
    for name in cpd_golden.group("estep"):
        c = cpd_golden.case("estep/" + name)
        es = obj.expectation_step(c["t_source"], c["target"], c["sigma2"], c["w"])
        assert np.max(np.abs(es.pt1 - c["pt1"])) < 2e-6, name
        assert rel_err(es.p1, c["p1"]) < 1e-5, name
        assert rel_err(es.px, c["px"]) < 1e-5, name
        assert abs(es.n_p - c["n_p"]) < 1e-6 * c["n_p"], name


def test_dead_column_is_zero(cpd_golden):
    from probreg_amd import cpd

    c = cpd_golden.case("estep/bunny_dead_column_w0")
    es = cpd.RigidCPD().expectation_step(c["t_source"], c["target"], c["sigma2"], c["w"])
    assert es.pt1[17] == 0.0  # cpd.py:81 - the far-away target point owns no probability mass
    assert abs(es.n_p - (c["target"].shape[0] - 1)) < 1e-4


def test_maximization_step_from_arrays(cpd_golden):
    """Public maximization_step(target, estep_res) signature with reference-produced EstepResults."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd

    c = cpd_golden.case("reg/synth_rigid_2k_k1")
    src, tgt = c["source"], c["target"]
    es = co.expectation_step(src @ np.eye(3), tgt, 0.05, 0.1)
    for kind, obj, direct in (
        ("rigid", cpd.RigidCPD(src), co.mstep_rigid(src, tgt, es)),
        ("rigid_noscale", cpd.RigidCPD(src, update_scale=False), co.mstep_rigid(src, tgt, es, False)),
        ("affine", cpd.AffineCPD(src), co.mstep_affine(src, tgt, es)),
    ):
        res = obj.maximization_step(tgt, cpd.EstepResult(*es))
        # the clouds are rounded to float32 on upload (after fp64 centring): ~1e-8 relative on the moments
        assert abs(res.sigma2 - direct.sigma2) < 1e-6 * direct.sigma2, kind
        assert abs(res.q - direct.q) < 1e-6 * abs(direct.q), kind
        if kind.startswith("rigid"):
            assert rel_err(res.transformation.rot, direct.params["rot"]) < 1e-6
            assert abs(res.transformation.scale - direct.params["scale"]) < 1e-6
        else:
            assert rel_err(res.transformation.b, direct.params["b"]) < 1e-6
        assert np.max(np.abs(res.transformation.t - direct.params["t"])) < 1e-6


def test_sigma2_init_vs_reference(cpd_golden):
    from probreg_amd import math_utils as mu

    m = cpd_golden.case("misc")
    assert abs(mu.squared_kernel_sum(m["x15"], m["x15"]) - m["sks_x15"]) < 1e-5  # tests/test_math_utils.py:7-11
    c = cpd_golden.case("reg/bunny_rigid_default")
    assert abs(mu.squared_kernel_sum(c["source"], c["target"]) - m["sks_bunny"]) < 1e-6 * m["sks_bunny"]


def test_rbf_kernel_vs_reference(cpd_golden):
    from probreg_amd import math_utils as mu

    m = cpd_golden.case("misc")
    g = mu.rbf_kernel(m["x15"] * 0.1, m["x15"] * 0.1, 1.0)
    assert np.allclose(g, g.T)  # tests/test_math_utils.py:13-16
    assert np.max(np.abs(g - m["rbf_x15_beta1"])) < 2e-7
    c = cpd_golden.case("reg/fish_rigid_default")
    g = mu.rbf_kernel(c["source"], c["source"], 2.0)
    assert np.max(np.abs(g - m["rbf_fish_beta2"])) < 2e-7


@pytest.mark.parametrize("n,m,k,w", [(6000, 6000, 6, 0.0), (5000, 3000, 5, 0.1)])
def test_rigid_and_affine_vs_oracle_medium(n, m, k, w):
    """Seeded synthetic clouds at a size the oracle finishes in seconds."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=11)
    p, s2, q, _ = co.registration("rigid", src, tgt, w=w, maxiter=k, tol=-1.0, closed_form_init=True, c_estep=n * m > 10 ** 7)
    res = cpd.registration_cpd(src, tgt, "rigid", w=w, maxiter=k, tol=-1.0)
    _check_rigid(res, p["rot"], p["t"], p["scale"], s2)
    src, tgt, _ = synthetic.affine_pair(n, m=m, seed=12)
    p, s2, q, _ = co.registration("affine", src, tgt, w=w, maxiter=k, tol=-1.0, closed_form_init=True, c_estep=n * m > 10 ** 7)
    res = cpd.registration_cpd(src, tgt, "affine", w=w, maxiter=k, tol=-1.0)
    _check_affine(res, p["b"], p["t"], s2)


def test_far_from_origin_clouds():
    """Coordinates with a large common offset (examples/face-x.txt style): centring in fp64 before
    the fp32 upload keeps parity (input conditioning)."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(1500, seed=21)
    off = np.array([1277.0, -350.0, 80.0])
    src, tgt = src + off, tgt + off
    p, s2, q, _ = co.registration("rigid", src, tgt, maxiter=6, tol=-1.0, closed_form_init=True)
    res = cpd.registration_cpd(src, tgt, "rigid", maxiter=6, tol=-1.0)
    assert rel_err(res.transformation.rot, p["rot"]) < TOL_TF
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2
    assert np.max(np.abs(res.transformation.t - p["t"])) < TOL_TF * np.max(np.abs(p["t"]))


def test_tf_init_params_and_callbacks():
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic, transformation as tf

    src, tgt, (r, t, s) = synthetic.rigid_pair(1200, seed=31)
    init = {"rot": synthetic.rot_zx(20.0, 5.0), "t": np.array([0.05, 0.0, 0.0])}
    p, s2, q, _ = co.registration("rigid", src, tgt, maxiter=4, tol=-1.0, tf_init_params=dict(init),
                                  closed_form_init=True)
    seen = []
    res = cpd.registration_cpd(src, tgt, "rigid", maxiter=4, tol=-1.0, tf_init_params=dict(init),
                               callbacks=[lambda tr: seen.append(tr)])
    assert len(seen) == 4 and all(isinstance(x, tf.RigidTransformation) for x in seen)
    _check_rigid(res, p["rot"], p["t"], p["scale"], s2)


def test_shards_sum_to_whole_on_one_gpu():
    """The all-reduce payload is additive over target shards: two half-target plans give the moments
    of the whole-target plan (this is the 8-GPU path of SURVEY.md 8e exercised on one device)."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    src, tgt, _ = synthetic.rigid_pair(5000, m=4000, seed=41)
    src32, tgt32 = (src - src.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
    params = np.zeros(_lib.PRG_NPARAMS)
    params[[0, 4, 8, 12]] = 1.0
    params[13] = 0.02

    def moments(t_local):
        plan = CpdPlan()
        plan.set_source(src32)
        plan.set_target(t_local, n_global=tgt32.shape[0])
        plan.set_params(params)
        plan.estep(0.1)
        m = plan.get_moments()
        plan.close()
        return m

    whole = moments(tgt32)
    parts = moments(tgt32[:2300]) + moments(tgt32[2300:])
    assert np.max(np.abs(whole[:23] - parts[:23])) <= 2e-6 * np.max(np.abs(whole[:23]))


def test_tuning_variants_agree():
    """Packed / scalar arithmetic, 2 / 4 points per lane and any segment count give the same moments."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    src, tgt, _ = synthetic.rigid_pair(7000, m=5000, seed=43)
    plan = CpdPlan()
    plan.set_source((src - src.mean(0)).astype(np.float32))
    plan.set_target((tgt - tgt.mean(0)).astype(np.float32))
    params = np.zeros(_lib.PRG_NPARAMS)
    params[[0, 4, 8, 12]] = 1.0
    params[13] = 0.01
    ref = None
    for rc, sc, rr, sr in [(2, 0, 2, 0), (4, 3, 4, 5), (-2, 1, -2, 1), (-4, 7, -4, 2), (2, 64, 2, 64)]:
        plan.set_tuning(rc, sc, rr, sr)
        plan.set_params(params)
        plan.estep(0.05)
        m = plan.get_moments()[:23]
        if ref is None:
            ref = m
        assert np.max(np.abs(m - ref)) <= 3e-6 * np.max(np.abs(ref)), (rc, sc, rr, sr)
    plan.close()


def test_full_size_identities():
    """BASELINE config C1 size (N = M = 100k): with w = 0 every column of P sums to one, hence
    n_p = N, sum_m px_m = sum_n x_n and sum_n pt1_n |x_n|^2 = sum_n |x_n|^2 (size-independent checks)."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    n = 100000
    src, tgt, _ = synthetic.rigid_pair(n, seed=0)
    s32, t32 = (src - src.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
    plan = CpdPlan()
    plan.set_source(s32)
    plan.set_target(t32)
    plan.init_sums()
    plan.init_params(None)
    for _ in range(3):
        plan.estep(0.0)
        mom = plan.get_moments()
        t64 = t32.astype(np.float64)
        assert abs(mom[0] - n) < 2e-6 * n
        assert np.max(np.abs(mom[1:4] - t64.sum(0))) < 2e-6 * n
        assert abs(mom[22] - np.sum(t64 * t64)) < 2e-6 * np.sum(t64 * t64)
        plan.mstep(_lib.PRG_TF_RIGID, True)
    p = plan.get_params()
    rot = p[:9].reshape(3, 3)
    assert np.allclose(rot @ rot.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(rot) - 1.0) < 1e-12
    plan.close()


def test_reference_style_random_rotation():
    """Port of the reference's own tests/test_cpd.py:9-22: recover a random rotation of a cloud
    (Euler angles to 1e-2, translation to 1e-4 - the reference's tolerances)."""
    from probreg_amd import cpd, synthetic

    rng = np.random.default_rng(5)
    pts = synthetic.surface(3000, 77) * 0.1
    ang = rng.uniform(0.0, np.pi / 4.0, 3)
    cx, cy, cz = np.cos(ang)
    sx, sy, sz = np.sin(ang)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    r = rz @ ry @ rx
    res = cpd.registration_cpd(pts, pts @ r.T)
    assert np.allclose(res.transformation.rot, r, atol=1e-2, rtol=1e-2)
    assert np.allclose(res.transformation.t, np.zeros(3), atol=1e-4, rtol=1e-4)


def test_error_behaviour():
    from probreg_amd import cpd

    x = np.random.default_rng(0).normal(size=(50, 3))
    with pytest.raises(ValueError):
        cpd.registration_cpd(x, x, "rigid", w=1.5)
    with pytest.raises(ValueError):
        cpd.registration_cpd(x, x[:, :2], "rigid")
    with pytest.raises(AssertionError):
        cpd.RigidCPD().expectation_step(x[0], x, 0.1)


def test_gauss_transform_direct():
    """reference gauss_transform.py:10-16 and tests/test_gauss_transform.py:17-28 (1e-4)."""
    from probreg_amd import gauss_transform as gt

    rng = np.random.default_rng(9)
    src = rng.uniform(size=(300, 3))
    tgt = rng.uniform(size=(170, 3))
    w = rng.uniform(size=300)
    for h in (1.0, 0.5, 0.05):
        want = np.array([np.dot(w, np.exp(-np.sum((t - src) ** 2, axis=1) / (h * h))) for t in tgt])
        got = gt.GaussTransform(src, h).compute(tgt, w)
        assert np.allclose(got, want, atol=1e-4, rtol=1e-4)
    w2 = rng.normal(size=(2, 300))
    got = gt.GaussTransform(src, 0.3).compute(tgt, w2)
    want = np.array([[np.dot(wr, np.exp(-np.sum((t - src) ** 2, axis=1) / 0.09)) for t in tgt] for wr in w2])
    assert got.shape == (2, 170) and np.allclose(got, want, atol=1e-4, rtol=1e-4)


# ---------------------------------------------------------------------------------------------
# non-rigid CPD (reference cpd.py:247-303)
# ---------------------------------------------------------------------------------------------
NONRIGID_CASES = ["bunny_nonrigid_default", "bunny_nonrigid_k5", "fish_nonrigid_default", "synth_nonrigid_1k_k1",
                  "synth_nonrigid_1k_k5"]


@pytest.mark.parametrize("name", NONRIGID_CASES)
def test_nonrigid_vs_reference(cpd_golden, name):
    """The recovered non-rigid transform is compared as T(Y) = Y + G W (what the reference's
    ``transformation.transform(source)`` returns).  W itself solves a system whose condition number is
    ~lambda_max(G)/(lmd sigma2) ~ 1e6..1e8, so 1-ulp differences in the float32 G (Eigen's vectorised
    expf in the reference vs a correctly rounded exp here) move W by up to ~1e-3 relative while T,
    sigma2 and q stay inside the north-star tolerance; W is checked at that looser level."""
    from probreg_amd import cpd

    c = cpd_golden.case("reg/" + name)
    niter = [0]
    res = cpd.registration_cpd(c["source"], c["target"], "nonrigid",
                               callbacks=[lambda t: niter.__setitem__(0, niter[0] + 1)], **_kwargs(c))
    want_sigma2, want_ts, want_w = c["out_sigma2"], c["out_tsource"], c["out_w"]
    if "default" in name:
        assert abs(niter[0] - c["out_niter"]) <= 2
        if niter[0] != c["out_niter"]:
            # stopped an iteration or two apart from the fp64 reference (the default test is absolute, |dq| < 1e-3 on
            # q = sigma2): compare the state at the iteration count the GPU reached, from the oracle
            from oracle import cpd_numpy as co

            p, s2, _, _ = co.registration("nonrigid", c["source"], c["target"], maxiter=niter[0], tol=-1.0)
            g = co.rbf_kernel(c["source"], c["source"], 2.0)
            want_sigma2, want_ts, want_w = s2, co.transform("nonrigid", p, c["source"], g), p["w"]
    assert abs(res.sigma2 - want_sigma2) <= TOL_SIGMA2 * want_sigma2
    ts = res.transformation.transform(c["source"])
    extent = np.max(np.abs(want_ts - want_ts.mean(0)))
    # bunny.pcd spans 0.08 units, so with beta = 2 every entry of G is within 2.5e-3 of 1.0: the float32
    # kernel carries ~15 significant bits of structure and the reference result itself moves by ~2e-4
    # relative under 1-ulp changes of G (numpy expf vs Eigen expf vs correctly rounded) - looser T there.
    # ([r5] granted 2e-4 - was 3e-4; the HIP path measures 1.3e-4, tools/slack_audit.py)
    tol_t = 2e-4 if name.startswith("bunny_nonrigid_k5") else TOL_TF
    assert np.max(np.abs(ts - want_ts)) < tol_t * extent
    wmax = np.max(np.abs(want_w))
    assert np.max(np.abs(res.transformation.w - want_w)) < 2e-2 * wmax


def test_nonrigid_g_matches_oracle(cpd_golden):
    from oracle import cpd_numpy as co
    from probreg_amd import cpd

    c = cpd_golden.case("reg/fish_nonrigid_default")
    reg = cpd.NonRigidCPD(c["source"], beta=2.0)
    g = reg._tf_obj.g
    want = co.rbf_kernel(c["source"], c["source"], 2.0)
    assert g.dtype == np.float32 and np.array_equal(g, g.T)
    assert np.max(np.abs(g - want)) <= 1.2e-7


def test_nonrigid_vs_oracle_medium():
    """M = 3000 (not a multiple of the 128 Cholesky block): seeded C3-style clouds, 4 fixed iterations."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt = synthetic.nonrigid_pair(3500, m=3000, seed=61)
    p, s2, q, _ = co.registration("nonrigid", src, tgt, maxiter=4, tol=-1.0, closed_form_init=True)
    g = co.rbf_kernel(src, src, 2.0)
    want = co.transform("nonrigid", p, src, g)
    res = cpd.registration_cpd(src, tgt, "nonrigid", maxiter=4, tol=-1.0)
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2
    got = res.transformation.transform(src)
    assert np.max(np.abs(got - want)) < TOL_TF * np.max(np.abs(want - want.mean(0)))


# ---------------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n", [(2, 3), (1, 5), (7, 1), (513, 1025)])
def test_tiny_and_odd_sizes_vs_oracle(m, n):
    """Sizes below one chunk / one wave and just past a block boundary."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd

    rng = np.random.default_rng(100 + m + n)
    src = rng.normal(size=(m, 3))
    tgt = rng.normal(size=(n, 3)) * 0.9 + 0.1
    want = co.expectation_step(src, tgt, 0.7, 0.2)
    got = cpd.RigidCPD().expectation_step(src, tgt, 0.7, 0.2)
    assert np.max(np.abs(got.pt1 - want.pt1)) < 2e-6
    assert rel_err(got.p1, want.p1) < 1e-5 and rel_err(got.px, want.px) < 1e-5


def test_two_dimensional_clouds_vs_oracle():
    """D = 2 (examples/cpd_affine2d.py): rigid and affine, fixed iterations."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd

    rng = np.random.default_rng(77)
    u = rng.uniform(0, 2 * np.pi, 800)
    src = np.stack([np.cos(u) * (1 + 0.3 * np.cos(3 * u)), 0.6 * np.sin(u) * (1 + 0.2 * np.sin(2 * u))], axis=1)
    th = 0.4
    r = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    tgt = src[rng.permutation(800)[:700]] @ r.T * 1.1 + np.array([0.2, -0.1]) + rng.normal(scale=0.01, size=(700, 2))
    for kind in ("rigid", "affine"):
        p, s2, q, _ = co.registration(kind, src, tgt, w=0.05, maxiter=8, tol=-1.0, closed_form_init=True)
        res = cpd.registration_cpd(src, tgt, kind, w=0.05, maxiter=8, tol=-1.0)
        lin = res.transformation.rot if kind == "rigid" else res.transformation.b
        assert lin.shape == (2, 2) and res.transformation.t.shape == (2,)
        assert rel_err(lin, p["rot"] if kind == "rigid" else p["b"]) < TOL_TF
        assert np.max(np.abs(res.transformation.t - p["t"])) < TOL_TF
        assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2


def test_duplicate_points_and_wide_kernel():
    """Exact duplicates in both clouds and a kernel much wider than the data (P nearly uniform)."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd

    rng = np.random.default_rng(5)
    base = rng.normal(size=(300, 3))
    src = np.concatenate([base, base[:50]])
    tgt = np.concatenate([base[::-1] + 0.01, base[:20] + 0.01])
    want = co.expectation_step(src, tgt, 50.0, 0.3)
    got = cpd.RigidCPD().expectation_step(src, tgt, 50.0, 0.3)
    assert rel_err(got.p1, want.p1) < 1e-5
    # px = sum_n P x_n nearly cancels here (uniform P, centred x): measure its error against p1 * |x|
    assert np.max(np.abs(got.px - want.px)) < 1e-5 * np.max(want.p1) * np.max(np.abs(tgt))
    assert np.max(np.abs(got.pt1 - want.pt1)) < 2e-6


def test_constrained_nonrigid_vs_reference():
    """ConstrainedNonRigidCPD (reference cpd.py:306-404): same solver with p1 + sigma2/alpha p1_tilde."""
    import os
    from conftest import GOLDEN_DIR, Golden
    from probreg_amd import cpd

    gold = Golden(os.path.join(GOLDEN_DIR, "cpd_constrained_golden.npz"))
    for name in gold.group("reg"):
        c = gold.case("reg/" + name)
        kw = {}
        if "arg_maxiter" in c:
            kw["maxiter"] = int(c["arg_maxiter"])
        if "arg_tol" in c:
            kw["tol"] = float(c["arg_tol"])
        niter = [0]
        res = cpd.registration_cpd(c["source"], c["target"], "nonrigid_constrained", alpha=float(c["alpha"]),
                                   idx_source=c["idx_source"], idx_target=c["idx_target"],
                                   callbacks=[lambda t: niter.__setitem__(0, niter[0] + 1)], **kw)
        assert abs(niter[0] - c["out_niter"]) <= 2, name
        if niter[0] != c["out_niter"]:  # compare at the iteration count the GPU reached (oracle, fixed count)
            from oracle import cpd_numpy as co

            p, s2, _, _ = co.registration("nonrigid_constrained", c["source"], c["target"], maxiter=niter[0], tol=-1.0,
                                          alpha=float(c["alpha"]), idx_source=c["idx_source"], idx_target=c["idx_target"])
            c = dict(c, out_sigma2=s2, out_tsource=co.transform("nonrigid", p, c["source"],
                                                              co.rbf_kernel(c["source"], c["source"], 2.0)))
        # With alpha = 1e-8 the prior rows of the system are weighted by sigma2/alpha ~ 1e7: a 1-ulp change of a
        # float32 G entry (6e-8) moves those rows by ~0.6 against c = lmd*sigma2 ~ 0.1, i.e. the reference's own
        # answer depends on how its expf rounds (numpy here, Eigen's vectorised expf in a real build).  Two steps of
        # fp64 iterative refinement in the solver do not move our result, so the residual gap is that input
        # sensitivity, not solver error; hold such cases to 2.5e-4 / 4e-4 instead of 1e-5 / 1e-4 ([r5]: were 5e-4 / 1e-3; the HIP
        # path measures 1.3e-4 / 2.2e-4, tools/slack_audit.py; tests/test_tolerance_justification.py shows the reference moving
        # by more than 1e-5 / 1e-4 under 1 ulp of G).
        loose = float(c["alpha"]) <= 1e-6
        assert abs(res.sigma2 - c["out_sigma2"]) <= (2.5e-4 if loose else TOL_SIGMA2) * c["out_sigma2"], name
        ts = res.transformation.transform(c["source"])
        extent = np.max(np.abs(c["out_tsource"] - c["out_tsource"].mean(0)))
        assert np.max(np.abs(ts - c["out_tsource"])) < (4e-4 if loose else TOL_TF) * extent, name
        # the constrained points are pulled onto their partners when alpha is tiny
        if float(c["alpha"]) <= 1e-6:
            d = ts[c["idx_source"]] - c["target"][c["idx_target"]]
            assert np.max(np.abs(d)) < 1e-3 * extent


# ---------------------------------------------------------------------------------------------
# exact culling: same numbers as the dense sweeps
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sigma2,w", [(3e-5, 0.0), (1e-3, 0.1), (0.05, 0.0)])
def test_culled_sweeps_equal_dense_sweeps(sigma2, w):
    """Default plan (Morton-sorted, culled) against a plan with sorting and culling switched off: the skipped
    (wave, group) blocks are exact zeros, so moments, pt1, p1 and px agree to float32 summation order."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    src, tgt, (r, t, _) = synthetic.rigid_pair(20000, m=15000, seed=51)
    z = (src @ r.T + t)                      # aligned clouds: small sigma2 is meaningful
    s32, t32 = (z - tgt.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
    params = np.zeros(_lib.PRG_NPARAMS)
    params[[0, 4, 8, 12]] = 1.0
    params[13] = sigma2
    out = []
    for opts in (dict(sort_source=True, sort_target=True, cull=True), dict(sort_source=False, sort_target=False, cull=False)):
        plan = CpdPlan()
        plan.set_options(**opts)
        plan.set_dense_engine(0)             # this test is about the CULLED VECTOR sweeps (tests/test_mfma_gpu.py has the others)
        plan.set_source(s32)
        plan.set_target(t32)
        plan.set_params(params)
        plan.estep(w)
        first = plan.get_moments()[:23]
        plan.estep(w)                        # second E-step: the column pass now runs with the seeded bound
        mom = plan.get_moments()[:23]
        pt1, p1, px = plan.get_estep()
        out.append((first, mom, pt1, p1, px))
        plan.close()
    (f0, m0, pt0, p0, x0), (f1, m1, pt1_, p1_, x1) = out
    scale = np.max(np.abs(m1))
    assert np.max(np.abs(f0 - f1)) <= 2e-6 * scale
    assert np.max(np.abs(m0 - m1)) <= 2e-6 * scale
    assert np.max(np.abs(pt0 - pt1_)) <= 2e-6
    assert np.max(np.abs(p0 - p1_)) <= 2e-6 * np.max(p1_)
    assert np.max(np.abs(x0 - x1)) <= 2e-6 * np.max(np.abs(x1))


@pytest.mark.parametrize("sigma2,w,m,n", [(1e-2, 0.0, 15000, 20000), (1e-3, 0.1, 20000, 9000), (5e-5, 0.0, 30011, 29989),
                                          (2e-4, 0.3, 700, 5000)])
def test_queue_sweeps_equal_grid_culled_sweeps(sigma2, w, m, n):
    """Sparse regime: the sweeps over the device-built work queue (default) evaluate the same (128 x 32) blocks with the same
    arithmetic as the grid of culled waves (prg_cpd_set_sparse_engine(0)); only the order in which a block's partial sums are
    combined differs (float32 / fp64 round-off).  The queue result does not depend on which wave popped which unit: two runs
    agree bit for bit."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    src, tgt, (r, t, _) = synthetic.rigid_pair(n, m=m, seed=61)
    z = (src @ r.T + t)
    s32, t32 = (z - tgt.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
    params = np.zeros(_lib.PRG_NPARAMS)
    params[[0, 4, 8, 12]] = 1.0
    params[13] = sigma2
    out = []
    for engine in (2, 0, 2):
        plan = CpdPlan()
        plan.set_dense_engine(0)
        plan.set_sparse_engine(engine)
        plan.set_source(s32)
        plan.set_target(t32)
        plan.set_params(params)
        plan.estep(w)
        first, pairs_first = plan.get_moments()[:23], plan.pair_counts()
        plan.estep(w)                        # second E-step: the column pass runs with the seeded bound
        out.append((first, plan.get_moments()[:23], plan.pair_counts(), pairs_first) + plan.get_estep())
        plan.close()
    q, g, q2 = out
    scale = np.max(np.abs(g[1]))
    assert np.max(np.abs(q[0] - g[0])) <= 2e-6 * scale and np.max(np.abs(q[1] - g[1])) <= 2e-6 * scale
    # the very same blocks were evaluated (the grid's unseeded first column pass also walks the pad groups of its last segments)
    assert q[2][1] == g[2][1] and q[3][1] == g[3][1] and g[2][0] >= q[2][0] >= 0.95 * g[2][0] and g[3][0] >= q[3][0] > 0, (q[2], g[2], q[3], g[3])
    assert np.max(np.abs(q[4] - g[4])) <= 2e-6 and np.max(np.abs(q[5] - g[5])) <= 2e-6 * np.max(g[5])
    assert np.max(np.abs(q[6] - g[6])) <= 2e-6 * np.max(np.abs(g[6]))
    for a, b in zip(q, q2):                  # reproducible to the bit
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_culled_registration_tracks_dense_registration():
    """30 EM iterations, culled vs dense, parameters compared every iteration (the seed of the column-pass bound
    comes from the previous iteration's minima plus the measured source motion)."""
    from probreg_amd import _lib, synthetic
    from probreg_amd.engine import CpdPlan

    src, tgt, _ = synthetic.rigid_pair(12000, seed=52)
    s32, t32 = (src - src.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
    plans = []
    for cull in (True, False):
        plan = CpdPlan()
        plan.set_options(sort_source=cull, sort_target=cull, cull=cull)
        plan.set_dense_engine(0)             # culled vs dense VECTOR sweeps
        plan.set_source(s32)
        plan.set_target(t32)
        plan.init_sums()
        plan.init_params(None)
        plans.append(plan)
    for it in range(30):
        ps = []
        for plan in plans:
            plan.estep(0.0)
            plan.mstep(_lib.PRG_TF_RIGID, True)
            ps.append(plan.get_params())
        assert np.max(np.abs(ps[0][:13] - ps[1][:13])) < 2e-6, it
        assert abs(ps[0][13] - ps[1][13]) <= 2e-6 * ps[1][13], it
    for plan in plans:
        plan.close()


def test_compute_l2_dist_vs_direct_formula():
    """reference cost_functions.py:33-41 (SVR / GMMReg objective + gradient) against the closed form."""
    from probreg_amd import cost_functions as cf

    rng = np.random.default_rng(21)
    mu_s, mu_t = rng.normal(size=(120, 3)), rng.normal(size=(90, 3)) * 1.2
    phi_s, phi_t = rng.uniform(0.5, 1.5, 120), rng.uniform(0.5, 1.5, 90)
    sigma = 0.4
    val, grad = cf.compute_l2_dist(mu_s, phi_s, mu_t, phi_t, sigma)
    z = (2.0 * np.pi * sigma ** 2) ** 1.5
    d2 = ((mu_s[:, None, :] - mu_t[None, :, :]) ** 2).sum(-1)
    k = np.exp(-d2 / (2.0 * sigma ** 2))          # h^2 = 2 sigma^2
    phi_j_e = k @ (phi_t / z)
    phi_mu = k @ (phi_t[:, None] * mu_t / z)
    want_val = -np.dot(phi_s, phi_j_e)
    want_grad = (phi_s[:, None] * phi_j_e[:, None] * mu_s - phi_s[:, None] * phi_mu) / (2.0 * sigma ** 2)
    assert abs(val - want_val) < 1e-5 * abs(want_val)
    assert np.max(np.abs(grad - want_grad)) < 1e-5 * np.max(np.abs(want_grad))


def test_single_rank_process_group_exercises_the_collective_path():
    """One-rank NCCL (RCCL) process group on this GPU: target 'sharding' over 1 rank, the moment block bound to a
    torch tensor and all-reduced every iteration on the plan's stream - the multi-GPU code path minus the peers."""
    import os
    import torch
    import torch.distributed as tdist
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    if tdist.is_initialized():
        pytest.skip("a process group already exists")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        src, tgt, _ = synthetic.rigid_pair(3000, m=2500, seed=71)
        res = cpd.registration_cpd(src, tgt, "rigid", w=0.1, maxiter=6, tol=-1.0)
        p, s2, q, _ = co.registration("rigid", src, tgt, w=0.1, maxiter=6, tol=-1.0, closed_form_init=True)
        _check_rigid(res, p["rot"], p["t"], p["scale"], s2)
        src, tgt = synthetic.nonrigid_pair(900, m=800, seed=72)
        res = cpd.registration_cpd(src, tgt, "nonrigid", maxiter=3, tol=-1.0)
        p, s2, q, _ = co.registration("nonrigid", src, tgt, maxiter=3, tol=-1.0, closed_form_init=True)
        assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2
    finally:
        tdist.destroy_process_group()


def test_far_from_origin_nonrigid_clouds():
    """Non-rigid CPD on clouds with a large common offset (advisor, round 1): the float32 upload happens in a frame
    shifted by the source's fp64 mean, so the E-step's differences and G keep their structure.  The problem is
    translation invariant, so the yardstick is the oracle on the SAME clouds without the offset (the reference's own
    float32 G of the offset coordinates has lost the fine structure and is no reference there)."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt = synthetic.nonrigid_pair(1300, m=1100, seed=31)
    off = np.array([1277.0, -350.0, 80.0])
    p, s2, q, _ = co.registration("nonrigid", src, tgt, maxiter=4, tol=-1.0, closed_form_init=True)
    want = co.transform("nonrigid", p, src, co.rbf_kernel(src, src, 2.0)) + off
    res = cpd.registration_cpd(src + off, tgt + off, "nonrigid", maxiter=4, tol=-1.0)
    assert abs(res.sigma2 - s2) <= 1e-4 * s2      # the shifted float32 coordinates are not bit-identical to the unshifted ones
    got = res.transformation.transform(src + off)
    assert np.max(np.abs(got - want)) < 2e-4 * np.max(np.abs(want - want.mean(0)))
    # public maximization_step in the same frame
    reg = cpd.NonRigidCPD(src + off)
    s2_0 = co.squared_kernel_sum_closed_form(src, tgt)
    es = reg.expectation_step(src + off, tgt + off, s2_0, 0.0)
    r1 = reg.maximization_step(tgt + off, es, s2_0)
    eo = co.expectation_step(src, tgt, s2_0, 0.0)
    po, s2o, _ = co.mstep_nonrigid(src, tgt, eo, s2_0, co.rbf_kernel(src, src, 2.0), 2.0)
    assert abs(r1.sigma2 - s2o) <= 1e-4 * s2o
