"""FilterReg's trajectory parity, the claim of DESIGN.md section 4 as tests (round-3 review, item 7).

The claim: with the splat in the reference's order the GPU E-step is the oracle's, bit for bit, ALONG WHOLE TRAJECTORIES; the
handful of `update_sigma2` trajectories that leave the 1e-4 tolerance do so because last-bit differences of the fp64 M-step sums
(block-wise on the GPU, numpy's pairwise order in the oracle: cc/kabsch.cc:13-45, filterreg.py:190-196) move a lattice cell
assignment a few iterations later - every single step, started from the oracle's state, stays well inside the tolerance.

Same 30 cases, generator and seed as tools/fuzz_filterreg.py / tools/fuzz_filterreg_hybrid.py (profiles/r4_fuzz_filterreg_hybrid.log).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

EXCURSIONS = (0, 4, 13, 19, 25)  # whole-trajectory G/G out of tolerance in profiles/r4_fuzz_filterreg_hybrid.log


@pytest.fixture(scope="module")
def hybrid():
    import fuzz_filterreg_hybrid as h
    from probreg_amd import _lib

    _lib.check(_lib.lib.prg_lattice_set_splat_mode(2))
    yield h
    _lib.check(_lib.lib.prg_lattice_set_splat_mode(1))


def test_gpu_estep_under_the_oracle_mstep_is_the_oracle_bit_for_bit(hybrid):
    """G/O (GPU lattice E-step + oracle M-step) against O/O over whole trajectories: identical bits in every case, the
    excursion cases included - so the E-step contributes nothing to them."""
    differing = []
    for c, src, tgt, kw, line in hybrid.fuzz_cases(30, 0):
        rot0, t0, s0 = hybrid.run("O/O", src, tgt, kw)
        rot, t, s2 = hybrid.run("G/O", src, tgt, kw)
        if not (np.array_equal(rot, rot0) and np.array_equal(t, t0) and s2 == s0):
            differing.append(line)
    assert not differing, differing


def test_every_single_step_from_the_oracle_state_is_within_tolerance(hybrid):
    """The excursion cases step by step: at each iteration the product does ONE EM iteration from the oracle's state
    (transform, sigma2) and is compared with the oracle's next state.  Bars: transform 2e-6 (rotation entries, translation over
    max(1, |t|)), sigma2 1e-5 relative, the north star itself (measured: 3.0e-7 and 7.6e-6 at worst) - against a whole-trajectory tolerance of 1e-4
    that the same cases miss by up to 0.7 when left to run freely."""
    from oracle import filterreg_numpy as fo
    from probreg_amd import filterreg

    worst = {}
    for c, src, tgt, kw, line in hybrid.fuzz_cases(30, 0):
        if c not in EXCURSIONS:
            continue
        assert kw["update_sigma2"]
        dim = src.shape[1]
        rot, t = np.identity(dim), np.zeros(dim)
        s2 = kw["sigma2"]
        e_tf = e_s2 = 0.0
        for _ in range(kw["maxiter"]):
            r1, t1, s1, _q, _ = fo.registration(src, tgt, sigma2=s2, update_sigma2=True, w=kw["w"], maxiter=1, tol=-1.0, rot0=rot,
                                               t0=t)
            res = filterreg.registration_filterreg(src, tgt, sigma2=s2, update_sigma2=True, w=kw["w"], maxiter=1, tol=-1.0,
                                                   tf_init_params={"rot": rot, "t": t})
            e_tf = max(e_tf, float(np.max(np.abs(res.transformation.rot - r1))),
                       float(np.max(np.abs(res.transformation.t - t1))) / max(1.0, float(np.max(np.abs(t1)))))
            e_s2 = max(e_s2, abs(res.sigma2 - s1) / abs(s1))
            rot, t, s2 = r1, t1, s1
        worst[c] = (e_tf, e_s2)
    assert set(worst) == set(EXCURSIONS)
    print("worst one-step differences (transform, sigma2):", worst)
    assert all(a < 2e-6 and b <= 1e-5 for a, b in worst.values()), worst   # (north star: sigma2 within 1e-5; measured worst 7.6e-6)
