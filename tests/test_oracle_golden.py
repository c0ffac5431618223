"""The numpy oracle (oracle/cpd_numpy.py) against fixtures produced by the reference's own code
(tests/golden/make_golden.py).  CPU only.  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import cpd_numpy as co
from conftest import rel_err

REG_CASES = [
    "bunny_rigid_default", "bunny_affine_default", "bunny_nonrigid_default", "bunny_nonrigid_k5",
    "bunny_rigid_noscale_w01_k10", "fish_rigid_default", "fish_affine_default", "fish_nonrigid_default",
    "synth_rigid_2k_k1", "synth_rigid_2k_k3", "synth_rigid_2k_k10", "synth_rigid_2k_w02_k5",
    "synth_affine_2k_k1", "synth_affine_2k_k10", "synth_nonrigid_1k_k1", "synth_nonrigid_1k_k5",
    "synth_rigid_ragged_k6",
]


def _kind(name):
    return "nonrigid" if "nonrigid" in name else ("affine" if "affine" in name else "rigid")


@pytest.mark.parametrize("name", REG_CASES)
def test_registration_matches_reference(cpd_golden, name):
    c = cpd_golden.case("reg/" + name)
    kind = _kind(name)
    kw = {}
    for k in ("w", "maxiter", "tol", "update_scale"):
        if "arg_" + k in c:
            kw[k] = c["arg_" + k]
    if "maxiter" in kw:
        kw["maxiter"] = int(kw["maxiter"])
    params, sigma2, q, niter = co.registration(kind, c["source"], c["target"], **kw)
    assert niter == c["out_niter"]
    # sigma2: the only difference is the fp32 summation order inside the sigma2 initialiser (~1e-8)
    assert abs(sigma2 - c["out_sigma2"]) <= 2e-7 * abs(c["out_sigma2"])
    if kind == "rigid":
        assert rel_err(params["rot"], c["out_rot"]) < 1e-7
        assert np.max(np.abs(params["t"] - c["out_t"])) < 1e-7
        assert abs(params["scale"] - c["out_scale"]) < 1e-7
    elif kind == "affine":
        assert rel_err(params["b"], c["out_b"]) < 1e-7
        assert np.max(np.abs(params["t"] - c["out_t"])) < 1e-7
    else:
        g = co.rbf_kernel(c["source"], c["source"], 2.0)
        ts = co.transform("nonrigid", params, c["source"], g)
        assert rel_err(ts, c["out_tsource"]) < 1e-6
    if c["out_sigma2"] > 2e-7:  # q is ill-conditioned once sigma2 sits on the eps32 clamp
        assert abs(q - c["out_q"]) <= 1e-6 * abs(c["out_q"])


@pytest.mark.parametrize("chunk", [64, 1024])
def test_estep_matches_reference(cpd_golden, chunk):
    for name in cpd_golden.group("estep"):
        c = cpd_golden.case("estep/" + name)
        es = co.expectation_step(c["t_source"], c["target"], c["sigma2"], c["w"], chunk=chunk)
        assert np.max(np.abs(es.pt1 - c["pt1"])) < 1e-12, name
        assert rel_err(es.p1, c["p1"]) < 1e-12, name
        assert rel_err(es.px, c["px"]) < 1e-12, name
        assert abs(es.n_p - c["n_p"]) < 1e-9, name


def test_unchunked_estep_matches_reference(cpd_golden):
    """The reference's own dense-matrix formulation restated line by line (what bench.py's cpu_baseline.reference_numpy
    times): equal to the reference's E-step outputs to 1e-12."""
    for name in cpd_golden.group("estep"):
        c = cpd_golden.case("estep/" + name)
        es = co.expectation_step_unchunked(c["t_source"], c["target"], c["sigma2"], c["w"])
        assert np.max(np.abs(es.pt1 - c["pt1"])) < 1e-12, name
        assert rel_err(es.p1, c["p1"]) < 1e-12, name
        assert rel_err(es.px, c["px"]) < 1e-12, name
        assert abs(es.n_p - c["n_p"]) < 1e-9, name


def test_dead_column_rule(cpd_golden):
    # cpd.py:81: a column whose every fp64 exp() underflowed gets den = eps32 -> P column == 0
    c = cpd_golden.case("estep/bunny_dead_column_w0")
    assert c["pt1"].min() == 0.0
    assert abs(c["n_p"] - (c["target"].shape[0] - 1)) < 1e-9


def test_math_utils_vectors(cpd_golden):
    m = cpd_golden.case("misc")
    x = m["x15"]
    # reference tests/test_math_utils.py:7-11
    brute = sum(np.sum((x[i] - x[j]) ** 2) for i in range(5) for j in range(5)) / (5 * 3 * 5)
    assert abs(m["sks_x15"] - brute) < 1e-6
    assert abs(co.squared_kernel_sum(x, x) - m["sks_x15"]) < 1e-6
    assert abs(co.squared_kernel_sum_closed_form(x, x) - m["sks_x15"]) < 1e-6
    g = co.rbf_kernel(x * 0.1, x * 0.1, 1.0)
    assert np.allclose(g, g.T)  # tests/test_math_utils.py:13-16
    assert np.max(np.abs(g - m["rbf_x15_beta1"])) < 1e-7


def test_moment_form_equals_direct_mstep(cpd_golden):
    c = cpd_golden.case("reg/synth_rigid_2k_k1")
    src, tgt = c["source"], c["target"]
    s2 = co.squared_kernel_sum(src, tgt)
    es = co.expectation_step(src, tgt, s2, 0.0)
    mom = co.moments_from_estep(src, tgt, es)
    for kind, direct in (("rigid", co.mstep_rigid(src, tgt, es)), ("affine", co.mstep_affine(src, tgt, es))):
        via = co.mstep_from_moments(kind, mom, 3)
        assert abs(via.sigma2 - direct.sigma2) < 1e-12 * abs(direct.sigma2) + 1e-15
        assert abs(via.q - direct.q) < 1e-9 * abs(direct.q)
        for k in direct.params:
            assert np.max(np.abs(np.asarray(via.params[k]) - np.asarray(direct.params[k]))) < 1e-10


def test_constrained_nonrigid_matches_reference():
    """ConstrainedNonRigidCPD (cpd.py:306-404) fixtures from the reference's own class."""
    import os
    from conftest import GOLDEN_DIR, Golden

    gold = Golden(os.path.join(GOLDEN_DIR, "cpd_constrained_golden.npz"))
    for name in gold.group("reg"):
        c = gold.case("reg/" + name)
        kw = {}
        if "arg_maxiter" in c:
            kw["maxiter"] = int(c["arg_maxiter"])
        if "arg_tol" in c:
            kw["tol"] = float(c["arg_tol"])
        params, sigma2, q, niter = co.registration("nonrigid_constrained", c["source"], c["target"], alpha=float(c["alpha"]),
                                                   idx_source=c["idx_source"], idx_target=c["idx_target"], **kw)
        assert niter == c["out_niter"], name
        assert abs(sigma2 - c["out_sigma2"]) <= 1e-6 * abs(c["out_sigma2"]), name
        g = co.rbf_kernel(c["source"], c["source"], 2.0)
        ts = co.transform("nonrigid", params, c["source"], g)
        assert rel_err(ts, c["out_tsource"]) < 1e-6, name


@pytest.mark.parametrize("name", ["fish_cold", "fish_warm3_w01", "synth_900_warm2", "fish_constrained"])
def test_nonrigid_mstep_restatement_matches_reference_single_call(mstep_golden, name):
    """oracle.cpd_numpy.mstep_nonrigid(_constrained) against one NonRigidCPD.maximization_step call of the reference
    on the reference's own E-step arrays (cpd.py:272-303, :377-404)."""
    c = mstep_golden.case("nonrigid/" + name)
    es = co.EstepResult(c["pt1"], c["p1"], c["px"], c["n_p"])
    g = co.rbf_kernel(c["source"], c["source"], float(c.get("ctor_beta", 2.0)))
    lmd = float(c.get("ctor_lmd", 2.0))
    if "ctor_idx_source" in c:
        m, dim = c["source"].shape
        p1_t, px_t = np.zeros(m), np.zeros((m, dim))
        p1_t[c["ctor_idx_source"]] = 1.0
        px_t[c["ctor_idx_source"]] = c["target"][c["ctor_idx_target"]]
        p, s2, q = co.mstep_nonrigid_constrained(c["source"], c["target"], es, c["sigma2_p"], g, lmd,
                                                 float(c["ctor_alpha"]), p1_t, px_t)
    else:
        p, s2, q = co.mstep_nonrigid(c["source"], c["target"], es, c["sigma2_p"], g, lmd)
    assert abs(s2 - c["out_sigma2"]) <= 1e-9 * c["out_sigma2"]
    assert np.max(np.abs(p["w"] - c["out_w"])) <= 1e-6 * np.max(np.abs(c["out_w"]))


@pytest.mark.parametrize("name", ["synth_pt2pt_update", "synth_pt2pt_fixed_w0", "fish2d_update", "synth_pt2pl_update"])
def test_filterreg_mstep_restatement_matches_reference_single_call(mstep_golden, name):
    """oracle.filterreg_numpy.maximization_step against one RigidFilterReg._maximization_step call of the reference
    (filterreg.py:158-196; Kabsch / twist solve restated from cc/kabsch.cc, cc/point_to_plane.cc)."""
    from oracle import filterreg_numpy as fo

    c = mstep_golden.case("filterreg/" + name)
    es = fo.EstepResult(c["m0"], c["m1"], c.get("m2"), c.get("nx"))
    res = fo.maximization_step(c["t_source"], c["target"], es, c["rot_p"], c["t_p"], c["sigma2"], w=c["w"],
                               objective_type="pt2pl" if "nx" in c else "pt2pt")
    assert np.max(np.abs(res.rot - c["out_rot"])) < 2e-6
    assert np.max(np.abs(res.t - c["out_t"])) < 2e-6
    assert abs(res.sigma2 - c["out_sigma2"]) <= 1e-7 * c["out_sigma2"]
    assert abs(res.q - c["out_q"]) <= 2e-6 * abs(c["out_q"])
