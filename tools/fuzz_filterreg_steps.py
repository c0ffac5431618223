#!/usr/bin/env python
"""Single-step parity of FilterReg along the ORACLE's trajectory: at every iteration the GPU starts from the oracle's
state (transform, sigma2) and does one EM iteration; its result is compared with the oracle's next state.  Separates
genuine differences from the chaotic amplification of 1e-7 changes of sigma2 through the lattice's cell assignment
(tests/test_tolerance_justification.py) that a whole-trajectory comparison on small clouds shows.

    python tools/fuzz_filterreg_steps.py m n dim w sigma2 iterations seed
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import filterreg_numpy as fo  # noqa: E402
from probreg_amd import filterreg, synthetic  # noqa: E402

m, n, dim = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w, sigma2, iters, seed = float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
src, tgt, _ = synthetic.filterreg_pair(n, m=m, seed=seed)
if dim == 2:
    src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
rot, t, s2 = np.identity(dim), np.zeros(dim), sigma2
worst = 0.0
for it in range(iters):
    r1, t1, s1, q1, _ = fo.registration(src, tgt, sigma2=s2, update_sigma2=True, w=w, maxiter=1, tol=-1.0, rot0=rot, t0=t,
                                        min_sigma2=0.0)
    res = filterreg.registration_filterreg(src, tgt, sigma2=s2, update_sigma2=True, w=w, maxiter=1, tol=-1.0, min_sigma2=0.0,
                                           tf_init_params={"rot": rot, "t": t})
    e = max(np.max(np.abs(res.transformation.rot - r1)), np.max(np.abs(res.transformation.t - t1)), abs(res.sigma2 - s1) / s1)
    worst = max(worst, e)
    print("it %2d sigma2 %.4e: one-step rot %.1e t %.1e sigma2 %.1e" % (
        it, s2, np.max(np.abs(res.transformation.rot - r1)), np.max(np.abs(res.transformation.t - t1)), abs(res.sigma2 - s1) / s1))
    rot, t, s2 = r1, t1, max(s1, 1e-4)
print("worst one-step difference %.2e" % worst)
