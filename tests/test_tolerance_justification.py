"""Every GPU parity test that grants more than the north-star tolerance (transform 1e-4, sigma2 1e-5) names a reason.
These CPU tests turn each reason into a measurement on the ORACLE (pinned to the reference): perturb the named input by
the named amount and the reference algorithm's own answer moves by at least the slack that was granted - so the slack
measures the problem's conditioning, not an error of the HIP path, and it cannot silently grow.

  bound granted                                         in                                    reason tested here
  T(Y) 3e-4 (bunny, beta = 2, 5 iterations)             test_cpd_gpu.py::test_nonrigid_vs_reference           1 ulp of float32 G
  sigma2 5e-4 / T(Y) 1e-3 (alpha <= 1e-6)               test_cpd_gpu.py::test_constrained_nonrigid_vs_reference  1 ulp of float32 G
  (none any more: pt2pl was granted 5e-4 / 2e-3 in round 1 on the grounds of float32 6 x 6 sums - measured below at
   ~1e-7, so the pt2pl tests now hold the north-star tolerances)
  1e-3 when iteration counts differ (FilterReg default) test_filterreg_gpu.py::test_registration_defaults_vs_reference  one EM iteration more or less
  sigma2 up to 2.5e-4 in 3 of 40 fuzzed FilterReg runs  DESIGN.md section 4                                   lattice cell flips amplify 1e-7 in sigma2
"""
import os

import numpy as np
import pytest

from oracle import cpd_numpy as co
from oracle import filterreg_numpy as fo
from conftest import GOLDEN_DIR, Golden


def _ulp_perturbed(g, seed, frac=1.0):
    """+-1 ulp on a random subset of the float32 entries (what a differently rounded expf does)."""
    rng = np.random.default_rng(seed)
    up = np.nextafter(g, np.float32(2.0))
    dn = np.nextafter(g, np.float32(-1.0))
    r = rng.random(g.shape)
    out = np.where(r < 0.5 * frac, up, np.where(r < frac, dn, g)).astype(np.float32)
    return out


def _nonrigid_iterations(src, tgt, g, k, beta=2.0, lmd=2.0, constrained=None):
    sigma2 = co.squared_kernel_sum(src, tgt)
    params = dict(w=np.zeros_like(src))
    for _ in range(k):
        ts = co.transform("nonrigid", params, src, g)
        es = co.expectation_step(ts, tgt, sigma2, 0.0)
        if constrained is None:
            params, sigma2, _ = co.mstep_nonrigid(src, tgt, es, sigma2, g, lmd)
        else:
            alpha, p1t, pxt = constrained
            params, sigma2, _ = co.mstep_nonrigid_constrained(src, tgt, es, sigma2, g, lmd, alpha, p1t, pxt)
    return co.transform("nonrigid", params, src, g), sigma2


def test_bunny_nonrigid_answer_moves_3e_4_under_one_ulp_of_g(cpd_golden):
    """bunny.pcd spans 0.08 units: with beta = 2 every entry of G is within 2.5e-3 of 1 and the system
    (diag(p1) G + lmd sigma2 I) W = ... is conditioned ~1e7.  Rounding the float32 kernel differently (+-1 ulp, as
    numpy's expf vs Eigen's vectorised expf vs a correctly rounded exp do) moves the reference's own T(Y) by more than
    1e-4 of the extent - the GPU test therefore grants 3e-4 there, and only there."""
    c = cpd_golden.case("reg/bunny_nonrigid_k5")
    src, tgt = c["source"], c["target"]
    g = co.rbf_kernel(src, src, 2.0)
    base, s2 = _nonrigid_iterations(src, tgt, g, 5)
    extent = np.max(np.abs(base - base.mean(0)))
    assert np.max(np.abs(base - c["out_tsource"])) < 1e-6 * extent  # the oracle reproduces the reference here
    moves = []
    for seed in (1, 2, 3):
        ts, s2p = _nonrigid_iterations(src, tgt, _ulp_perturbed(g, seed), 5)
        moves.append(np.max(np.abs(ts - base)) / extent)
    assert max(moves) > 1.0e-4, moves     # the reference itself is not determined to 1e-4 here ...
    assert max(moves) < 1.0e-3, moves     # ... but to well under the 3e-4 x a few the test grants


def test_well_scaled_nonrigid_answer_does_not_move_under_one_ulp_of_g():
    """Counter-check: on the C3-style synthetic cloud (extent ~2.4, G entries spread over (0, 1]) the same perturbation
    stays far inside 1e-4 - which is why every other non-rigid test keeps the north-star tolerance."""
    from probreg_amd import synthetic

    src, tgt = synthetic.nonrigid_pair(700, m=600, seed=5)
    g = co.rbf_kernel(src, src, 2.0)
    base, s2 = _nonrigid_iterations(src, tgt, g, 4)
    ts, s2p = _nonrigid_iterations(src, tgt, _ulp_perturbed(g, 7), 4)
    extent = np.max(np.abs(base - base.mean(0)))
    assert np.max(np.abs(ts - base)) < 2e-5 * extent
    assert abs(s2p - s2) < 2e-6 * s2


def test_constrained_answer_moves_under_one_ulp_of_g_when_alpha_is_tiny():
    """alpha = 1e-8 weights the prior rows by sigma2 / alpha ~ 1e6..1e7: the reference's sigma2 and T(Y) move by more
    than the north-star tolerances when G is rounded differently, by less than the 5e-4 / 1e-3 the GPU test grants."""
    gold = Golden(os.path.join(GOLDEN_DIR, "cpd_constrained_golden.npz"))
    c = gold.case("reg/fish_alpha1e-8_k6")
    src, tgt = c["source"], c["target"]
    m, dim = src.shape
    pairs = np.unique(np.stack([c["idx_source"], c["idx_target"]], axis=1), axis=0)
    p1t, pxt = np.zeros(m), np.zeros((m, dim))
    np.add.at(p1t, pairs[:, 0], 1.0)
    np.add.at(pxt, pairs[:, 0], tgt[pairs[:, 1]])
    g = co.rbf_kernel(src, src, 2.0)
    base, s2 = _nonrigid_iterations(src, tgt, g, 6, constrained=(float(c["alpha"]), p1t, pxt))
    extent = np.max(np.abs(base - base.mean(0)))
    assert abs(s2 - c["out_sigma2"]) < 1e-6 * s2
    worst_t, worst_s = 0.0, 0.0
    for seed in (1, 2, 3):
        ts, s2p = _nonrigid_iterations(src, tgt, _ulp_perturbed(g, seed), 6, constrained=(float(c["alpha"]), p1t, pxt))
        worst_t = max(worst_t, np.max(np.abs(ts - base)) / extent)
        worst_s = max(worst_s, abs(s2p - s2) / s2)
    assert worst_s > 1e-5 or worst_t > 1e-4, (worst_s, worst_t)   # beyond the north-star tolerance on its own
    assert worst_s < 5e-4 and worst_t < 1e-3, (worst_s, worst_t)  # inside what the GPU test grants


def test_pt2pl_float32_normal_equations_do_not_justify_any_slack(fr_golden):
    """cc/point_to_plane.cc accumulates the 6 x 6 system in float32, the HIP path in fp64.  Round 1 granted the pt2pl tests
    5e-4 on that ground; measured on the golden case, eight iterations with float32 vs float64 normal equations end
    < 1e-6 apart in the transform (sigma2 4e-6, q 4e-6) - so no slack is justified and the GPU tests hold the north-star tolerances (the HIP path is 4e-8 /
    8e-7 from the reference there, tools/slack_audit.py)."""
    c = fr_golden.case("pt2pl/pt2pl_synth_update_k8")
    src, tgt, nrm = c["source"], c["target"], c["normals"]
    sigma2 = float(c["arg_sigma2"])
    r32, t32, s32, q32r, _ = fo.registration(src, tgt, sigma2=sigma2, update_sigma2=True, maxiter=8, tol=-1.0,
                                             target_normals=nrm, objective_type="pt2pl")
    assert np.max(np.abs(r32 - c["out_rot"])) < 1e-6   # the oracle follows the reference
    orig = fo.pt2pl_f32

    def pt2pl_f64(model, target, normal, weight):
        vv, tt, nn, ww = (np.asarray(a, dtype=np.float64) for a in (model, target, normal, weight))
        rr = np.einsum("kd,kd->k", nn, tt - vv)
        jj = np.concatenate([np.cross(vv, nn), nn], axis=1)
        return (np.linalg.solve(np.einsum("k,ki,kj->ij", ww, jj, jj), np.einsum("k,k,ki->i", ww, rr, jj)),
                np.sum(ww * ww * rr * rr))

    fo.pt2pl_f32 = pt2pl_f64
    try:
        r64, t64, s64, q64r, _ = fo.registration(src, tgt, sigma2=sigma2, update_sigma2=True, maxiter=8, tol=-1.0,
                                                 target_normals=nrm, objective_type="pt2pl")
    finally:
        fo.pt2pl_f32 = orig
    assert max(np.max(np.abs(r64 - r32)), np.max(np.abs(t64 - t32))) < 1e-6
    assert abs(q64r - q32r) / abs(q64r) < 2e-5 and abs(s64 - s32) / s64 < 1e-5


def test_filterreg_default_run_moves_1e_3_per_iteration_near_its_stopping_point(fr_golden):
    """The default driver stops on |q - q_prev| < 1e-3 (an ABSOLUTE test on q ~ 3e2): float32 lattice noise can make
    the HIP path stop one or two iterations apart from the reference.  One more iteration of the reference itself still
    moves the transform by ~1e-3 at that point - the bound granted when (and only when) the iteration counts differ."""
    c = fr_golden.case("reg/bunny_default")
    src, tgt = c["source"], c["target"]
    k = int(c["out_niter"])
    ra, ta, _, _, na = fo.registration(src, tgt, maxiter=k, tol=-1.0)
    rb, tb, _, _, nb = fo.registration(src, tgt, maxiter=k + 2, tol=-1.0)
    assert np.max(np.abs(ra - c["out_rot"])) < 1e-5       # the oracle follows the reference to its stopping point
    step = max(np.max(np.abs(ra - rb)), np.max(np.abs(ta - tb)))
    assert 1e-5 < step < 1e-3, step


def test_filterreg_sigma2_trajectory_amplifies_1e_7_through_the_lattice():
    """Cell assignment on the permutohedral lattice is a rounding operation of position / sigma: a 1e-7 relative change
    of sigma2 (float32 summation order in one M-step) moves points across cell borders and comes back 2-16 x larger
    from the next E/M step; over a sigma2-updating registration of a small cloud the REFERENCE ALGORITHM ITSELF ends
    1e-5 .. 1e-4 away from its own unperturbed run.  That is the whole story of the three fuzzed FilterReg
    configurations (of 40) whose sigma2 ends up to 2.5e-4 from the reference; the fixtures the
    GPU tests hold to 1e-5 are larger clouds / shorter runs where the amplification stays below it."""
    r = np.array([[np.cos(0.2), -np.sin(0.2), 0.0], [np.sin(0.2), np.cos(0.2), 0.0], [0.0, 0.0, 1.0]])
    worst = 0.0
    for seed in (5, 6, 7):
        rng = np.random.default_rng(seed)
        src = rng.uniform(-1.0, 1.0, (300, 3))
        tgt = src[rng.permutation(300)[:260]] @ r.T + rng.normal(0.0, 0.02, (260, 3))
        base = fo.registration(src, tgt, sigma2=0.05, update_sigma2=True, w=0.05, maxiter=12, tol=-1.0)[2]
        for eps in (1e-7, -1e-7):
            s2 = fo.registration(src, tgt, sigma2=0.05 * (1.0 + eps), update_sigma2=True, w=0.05, maxiter=12, tol=-1.0)[2]
            worst = max(worst, abs(s2 - base) / base)
    assert worst > 1e-5, worst
