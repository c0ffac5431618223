"""The sparse-regime work queue (csrc/cpd_sweeps_queue.hip) behind the PUBLIC API: by default it only serves clouds of
>= 32768 points on both sides, so the small-cloud tests never reach it.  ``PRG_SPARSE_ENGINE=2`` forces it for every plan created
while it is set; each case below runs the same registration / E-step with the queue forced and with it switched off (the grid
of culled waves) and holds the two to each other - both evaluate exactly the same pairs - and, where a fixture exists, to the
reference's own output.  Covers 2-D clouds, ragged and tiny sizes (fewer points than one block of 128), outlier weights,
per-source weights (the BCPD E-step), affine and non-rigid drivers.  Reference: probreg/cpd.py:71-88 (E-step), bcpd.py:53-72."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, Golden, rel_err

pytestmark = pytest.mark.gpu


def _both_engines(monkeypatch, fn):
    out = []
    for mode in ("2", "0"):
        monkeypatch.setenv("PRG_SPARSE_ENGINE", mode)
        monkeypatch.setenv("PRG_DENSE_ENGINE", "0")  # the vector-pipe sweeps from the first E-step on
        out.append(fn())
    return out


@pytest.mark.parametrize("kind,m,n,dim,w,iters", [("rigid", 3000, 2500, 3, 0.0, 30), ("rigid", 900, 4000, 2, 0.2, 25),
                                                  ("affine", 2100, 2100, 3, 0.1, 30), ("rigid", 77, 50, 3, 0.0, 12),
                                                  ("affine", 130, 5000, 2, 0.0, 20), ("nonrigid", 1500, 1300, 3, 0.0, 12)])
def test_registrations_agree_between_queue_and_grid(monkeypatch, kind, m, n, dim, w, iters):
    from probreg_amd import cpd, synthetic

    if kind == "nonrigid":
        src, tgt = synthetic.nonrigid_pair(n, m=m, seed=7)
    else:
        src, tgt, _ = (synthetic.rigid_pair if kind == "rigid" else synthetic.affine_pair)(n, m=m, seed=7)
    if dim == 2:
        src, tgt = src[:, :2].copy(), tgt[:, :2].copy()

    def run():
        res = cpd.registration_cpd(src, tgt, kind, w=w, maxiter=iters, tol=-1.0)
        return res.sigma2, res.q, res.transformation.transform(src)

    (s2q, qq, tq), (s2g, qg, tg) = _both_engines(monkeypatch, run)
    assert abs(s2q - s2g) <= 2e-6 * s2g
    assert abs(qq - qg) <= 2e-6 * abs(qg)
    assert np.max(np.abs(tq - tg)) <= 2e-6 * max(1.0, float(np.max(np.abs(tg))))


def test_late_regime_against_the_oracle_through_the_queue(monkeypatch):
    """sigma2 down at the noise level (most blocks culled): the queue against the CPU oracle, not only against the grid."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    monkeypatch.setenv("PRG_SPARSE_ENGINE", "2")
    src, tgt, _ = synthetic.rigid_pair(6000, m=5000, seed=17)
    res = cpd.registration_cpd(src, tgt, "rigid", w=0.05, maxiter=35, tol=-1.0)
    p, s2, q, _ = co.registration("rigid", src, tgt, w=0.05, maxiter=35, tol=-1.0, closed_form_init=True, c_estep=True)
    assert rel_err(res.transformation.rot, p["rot"]) < 1e-4 and np.max(np.abs(res.transformation.t - p["t"])) < 1e-4
    assert abs(res.sigma2 - s2) <= 1e-5 * s2


@pytest.mark.parametrize("name", ["alpha_vec_w0.1", "small_sigma2_w0.3", "planar_w0.05"])
def test_weighted_estep_through_the_queue(monkeypatch, name):
    """Per-source weights ride in z4.w as an additive squared distance; the queue's column pass runs unseeded for them."""
    from probreg_amd import bcpd

    c = Golden(os.path.join(GOLDEN_DIR, "bcpd_golden.npz")).case("estep/" + name)

    def run():
        reg = bcpd.CombinedBCPD(c["t_source"])
        es = reg.expectation_step(c["t_source"], c["target"], c["scale"], c["alpha"], c["sigma_diag"], c["sigma2"], c["w"])
        return es.nu, es.nu_d, es.px

    q, g = _both_engines(monkeypatch, run)
    for a, b in zip(q, g):
        assert np.max(np.abs(a - b)) <= 2e-6 * np.max(np.abs(b))
    assert rel_err(q[0], c["out_nu"]) < 2e-5 and rel_err(q[2], c["out_px"]) < 2e-5
