#!/usr/bin/env python
"""Randomised parity sweep for FilterReg (rigid, point-to-point): GPU against the oracle (C restatement of the
lattice + float32 Kabsch) over random sizes, dimensions, outlier weights, sigma2 policies and iteration counts.

    python tools/fuzz_filterreg.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import filterreg_numpy as fo  # noqa: E402
from probreg_amd import filterreg, synthetic  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad, worst, t0 = 0, 0.0, time.time()
    for c in range(cases):
        m = int(rng.choice([rng.integers(5, 60), rng.integers(60, 800), rng.integers(800, 5000)]))
        n = int(rng.choice([rng.integers(5, 60), rng.integers(60, 800), rng.integers(800, 5000)]))
        dim = int(rng.choice([2, 3]))
        w = float(rng.choice([0.0, 0.05, 0.3]))
        upd = bool(rng.integers(0, 2))
        iters = int(rng.integers(1, 14))
        sigma2 = None if rng.random() < 0.5 else float(10 ** rng.uniform(-3.5, -1.0))
        seed = int(rng.integers(0, 10 ** 6))
        src, tgt, _ = synthetic.filterreg_pair(n, m=m, seed=seed)
        if dim == 2:
            src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
        kw = dict(sigma2=sigma2, update_sigma2=upd, w=w, maxiter=iters, tol=-1.0)
        res = filterreg.registration_filterreg(src, tgt, **kw)
        rot, t, s2, q, _ = fo.registration(src, tgt, **kw)
        e_r = float(np.max(np.abs(res.transformation.rot - rot)))
        e_t = float(np.max(np.abs(res.transformation.t - t))) / max(1.0, float(np.max(np.abs(t))))
        e_s = abs(res.sigma2 - s2) / max(abs(s2), 1e-300)
        err = max(e_r, e_t, 10.0 * e_s)
        worst = max(worst, err)
        flag = "" if err < 1e-4 else "   <-- OUT OF TOLERANCE"
        bad += bool(flag)
        print("case %2d m=%4d n=%4d dim=%d w=%.2f update=%d sigma2=%s it=%2d seed=%6d: rot %.1e t %.1e sigma2 %.1e%s" % (
            c, m, n, dim, w, upd, "auto" if sigma2 is None else "%.1e" % sigma2, iters, seed, e_r, e_t, e_s, flag))
    print("%d cases, %d out of tolerance, worst %.2e, %.0f s" % (cases, bad, worst, time.time() - t0))


if __name__ == "__main__":
    main()
