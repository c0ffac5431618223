#!/usr/bin/env python
"""Wall time of whole registrations through the public API (host convergence test every iteration).

    python tools/time_registration.py [n]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import cpd, filterreg, synthetic  # noqa: E402


def timed(label, fn, reps=2):
    for r in range(reps):
        it = [0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fn(lambda tr: it.__setitem__(0, it[0] + 1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%-44s %4d iterations  %8.1f ms  (%.3f ms / iteration)  sigma2=%.4e" % (label, it[0], dt * 1e3, dt * 1e3 / max(it[0], 1),
                                                                             res.sigma2))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    src, tgt, _ = synthetic.rigid_pair(n, seed=0)
    timed("registration_cpd rigid N=M=%d tol=1e-3" % n, lambda cb: cpd.registration_cpd(src, tgt, "rigid", callbacks=[cb]))
    timed("registration_cpd rigid N=M=%d 100 it" % n,
          lambda cb: cpd.registration_cpd(src, tgt, "rigid", maxiter=100, tol=-1.0, callbacks=[cb]))
    src, tgt, _ = synthetic.affine_pair(n, seed=0)
    timed("registration_cpd affine N=M=%d tol=1e-3" % n, lambda cb: cpd.registration_cpd(src, tgt, "affine", callbacks=[cb]))
    src, tgt, _ = synthetic.filterreg_pair(n, seed=0)
    timed("registration_filterreg N=M=%d tol=1e-3" % n,
          lambda cb: filterreg.registration_filterreg(src, tgt, update_sigma2=True, w=0.05, callbacks=[cb]))


if __name__ == "__main__":
    main()
