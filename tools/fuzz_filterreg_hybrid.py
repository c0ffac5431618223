#!/usr/bin/env python
"""WHERE do the FilterReg `update_sigma2` trajectories that leave the tolerance come from - E-step or M-step?  Tested, not
argued: the 30 fuzz cases of tools/fuzz_filterreg.py (same generator, same seed) are run through four drivers that all
follow the reference's loop (filterreg.py:120-147, restated in oracle/filterreg_numpy.py) and differ only in who computes what:

  O/O   oracle lattice E-step + oracle M-step                        (the yardstick)
  G/O   GPU lattice E-step (reference-order splat, prg_lattice_set_splat_mode(2)) + oracle M-step
  O/G   oracle lattice E-step + GPU M-step (RigidFilterReg._maximization_step on explicit arrays)
  G/G   the product (registration_filterreg), reference-order splat

If the E-step is bit-identical along the whole trajectory, G/O must equal O/O BIT FOR BIT in every case; whatever
excursions remain in G/G must then show up in O/G - the fp64 M-step sums (block-wise on the GPU, numpy's pairwise order in
the oracle) - and nowhere else.          usage: fuzz_filterreg_hybrid.py [cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import filterreg_numpy as fo  # noqa: E402
from oracle import permutohedral as oph  # noqa: E402
from probreg_amd import _lib, filterreg, gaussian_filtering as gf, synthetic, transformation as tf  # noqa: E402


class GpuLattice(object):
    """oracle.permutohedral.Lattice's interface on the product's lattice (what the G/O driver plugs into the oracle)."""

    def __init__(self, points, with_blur=True, prefer_ref=False):
        self._p = gf.Permutohedral(np.asarray(points, dtype=np.float32), with_blur)
        self.lattice_size = self._p.get_lattice_size()

    def filter(self, values):
        return self._p.filter(np.asarray(values, dtype=np.float32))


def gpu_mstep(t_source, target, es, rot_p, t_p, sigma2, w=0.0, objective_type="pt2pt"):
    res = filterreg.RigidFilterReg._maximization_step(t_source, target, filterreg.EstepResult(es.m0, es.m1, es.m2, es.nx),
                                                      tf.RigidTransformation(rot_p, t_p), sigma2, w, objective_type)
    if res.q is None:
        return fo.MstepResult(rot_p, t_p, sigma2, None)
    return fo.MstepResult(res.transformation.rot, res.transformation.t, res.sigma2, res.q)


def run(driver, src, tgt, kw):
    lat, mstep = oph.Lattice, fo.maximization_step
    if driver == "G/G":
        res = filterreg.registration_filterreg(src, tgt, **kw)
        return res.transformation.rot, res.transformation.t, res.sigma2
    if driver[0] == "G":
        lat = GpuLattice
    if driver[2] == "G":
        mstep = gpu_mstep
    keep = fo.ph.Lattice, fo.maximization_step
    fo.ph.Lattice, fo.maximization_step = lat, mstep
    try:
        rot, t, s2, _q, _ = fo.registration(src, tgt, **kw)
    finally:
        fo.ph.Lattice, fo.maximization_step = keep
    return rot, t, s2


def fuzz_cases(cases=30, seed=0):
    """The generator of tools/fuzz_filterreg.py: yields (index, source, target, keyword arguments, description)."""
    rng = np.random.default_rng(seed)
    for c in range(cases):
        m = int(rng.choice([rng.integers(5, 60), rng.integers(60, 800), rng.integers(800, 5000)]))
        n = int(rng.choice([rng.integers(5, 60), rng.integers(60, 800), rng.integers(800, 5000)]))
        dim = int(rng.choice([2, 3]))
        w = float(rng.choice([0.0, 0.05, 0.3]))
        upd = bool(rng.integers(0, 2))
        iters = int(rng.integers(1, 14))
        sigma2 = None if rng.random() < 0.5 else float(10 ** rng.uniform(-3.5, -1.0))
        seed_c = int(rng.integers(0, 10 ** 6))
        src, tgt, _ = synthetic.filterreg_pair(n, m=m, seed=seed_c)
        if dim == 2:
            src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
        kw = dict(sigma2=sigma2, update_sigma2=upd, w=w, maxiter=iters, tol=-1.0)
        yield c, src, tgt, kw, "case %2d m=%4d n=%4d dim=%d w=%.2f update=%d it=%2d:" % (c, m, n, dim, w, upd, iters)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    _lib.check(_lib.lib.prg_lattice_set_splat_mode(2))
    drivers = ("G/O", "O/G", "G/G")
    bad = {d: 0 for d in drivers}
    bitwise = 0
    print("# error against O/O: max(|d rot|, |d t| / max(1, |t|), 10 x relative sigma2 error); tolerance 1e-4")
    for c, src, tgt, kw, line in fuzz_cases(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0):
        rot0, t0, s0 = run("O/O", src, tgt, kw)
        for d in drivers:
            rot, t, s2 = run(d, src, tgt, kw)
            err = max(float(np.max(np.abs(rot - rot0))), float(np.max(np.abs(t - t0))) / max(1.0, float(np.max(np.abs(t0)))),
                      10.0 * abs(s2 - s0) / max(abs(s0), 1e-300))
            same = np.array_equal(rot, rot0) and np.array_equal(t, t0) and s2 == s0
            if d == "G/O":
                bitwise += bool(same)
            bad[d] += err >= 1e-4
            line += "  %s %s" % (d, "bit-equal" if same else "%.1e%s" % (err, " OUT" if err >= 1e-4 else ""))
        print(line)
    print("%d cases: G/O bit-equal to O/O in %d; out of tolerance: G/O %d, O/G %d, G/G %d" % (cases, bitwise, bad["G/O"], bad["O/G"],
                                                                                        bad["G/G"]))


if __name__ == "__main__":
    main()
