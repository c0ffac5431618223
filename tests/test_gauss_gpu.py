"""SURVEY.md 8 rows a11 / f3 on the GPU against the REFERENCE's own outputs (tests/golden/gauss_golden.npz, written by
tests/golden/make_golden.py gauss from probreg/gauss_transform.py and probreg/cost_functions.py run unmodified):
``GaussTransform.compute`` with 1-D / 2-D / default weights, bandwidths below and above the reference's ``sw_h`` switch,
5000 x 5000 points, 2-D clouds, clouds far from the origin; ``compute_l2_dist`` value and gradient.

Tolerance: 1e-5 of the largest |output| of the case (the GPU forms the differences in fp64 and evaluates the squared
distance and the exponential in fp32: a term's relative error is ~ (d / h)^2 x 1.2e-7, sums are fp64; v_exp_f32 flushes
terms below 2^-126 of the largest, which 1e-5 of the largest |output| does not see)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, Golden

pytestmark = pytest.mark.gpu

TOL = 1e-5
G = Golden(os.path.join(GOLDEN_DIR, "gauss_golden.npz"))


@pytest.mark.parametrize("name", G.group("gt"))
def test_gauss_transform_vs_reference(name):
    from probreg_amd import gauss_transform as gt

    c = G.case("gt/" + name)
    got = gt.GaussTransform(c["source"], c["h"]).compute(c["target"], c.get("weights"))
    want = c["out"]
    assert got.shape == want.shape
    scale = np.max(np.abs(want))
    # fp32 exponentials: a term below 2^-126 is flushed to zero (the reference's fp64 keeps it down to 1e-308) - an absolute
    # floor of 2^-126 x sum |w|; only the `..._far` case (every exponent below -180) lives there
    w = c.get("weights")
    floor = 2.0 ** -126 * (np.sum(np.abs(w)) if w is not None else c["source"].shape[0])
    assert np.max(np.abs(got - want)) <= TOL * scale + floor, (name, np.max(np.abs(got - want)) / scale)
    # every entry that matters individually as well: relative 1e-4 wherever |want| is above 1e-3 of the largest
    big = (np.abs(want) > 1e-3 * scale) & (np.abs(want) > 1e3 * floor)
    if np.any(big):
        assert np.max(np.abs(got[big] - want[big]) / np.abs(want[big])) <= 1e-4, name


@pytest.mark.parametrize("name", G.group("l2"))
def test_compute_l2_dist_vs_reference(name):
    from probreg_amd import cost_functions as cf

    c = G.case("l2/" + name)
    f, g = cf.compute_l2_dist(c["mu_source"], c["phi_source"], c["mu_target"], c["phi_target"], c["sigma"])
    assert abs(f - c["out_f"]) <= TOL * abs(c["out_f"]), (name, f, c["out_f"])
    assert g.shape == c["out_g"].shape
    assert np.max(np.abs(g - c["out_g"])) <= TOL * np.max(np.abs(c["out_g"])), name


def test_weight_rows_share_sweeps_and_match_single_rows():
    """Seven weight rows = one sweep of four, one of two, one of one: each row must equal its own single-row call bit for bit
    (the sums are fp64 fma chains over the same source order whatever the grouping)."""
    from probreg_amd import gauss_transform as gt

    rng = np.random.default_rng(3)
    src, tgt = rng.normal(size=(1000, 3)), rng.normal(size=(777, 3))
    w = rng.normal(size=(7, 1000))
    tr = gt.GaussTransform(src, 0.7)
    all_rows = tr.compute(tgt, w)
    assert all_rows.shape == (7, 777)
    for k in range(7):
        assert np.array_equal(all_rows[k], tr.compute(tgt, w[k]))
    with pytest.raises(ValueError):
        tr.compute(tgt, np.zeros((2, 2, 1000)))
