#!/usr/bin/env python
"""Randomised parity sweep for NonRigidCPD: GPU against the numpy oracle (transformed source and sigma2).

    python tools/fuzz_nonrigid.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpd_numpy as co  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad, worst, t0 = 0, 0.0, time.time()
    for c in range(cases):
        m = int(rng.choice([rng.integers(5, 130), rng.integers(130, 700), rng.integers(700, 1800)]))
        n = int(rng.choice([rng.integers(5, 130), rng.integers(130, 700), rng.integers(700, 2500)]))
        dim = int(rng.choice([2, 3]))
        w = float(rng.choice([0.0, 0.1, 0.5]))
        beta = float(rng.choice([0.5, 2.0, 5.0]))
        lmd = float(rng.choice([0.5, 2.0, 10.0]))
        iters = int(rng.integers(1, 10))
        seed = int(rng.integers(0, 10 ** 6))
        src, tgt = synthetic.nonrigid_pair(n, m=m, seed=seed)
        if dim == 2:
            src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
        res = cpd.registration_cpd(src, tgt, "nonrigid", w=w, maxiter=iters, tol=-1.0, beta=beta, lmd=lmd)
        p, s2, q, _ = co.registration("nonrigid", src, tgt, w=w, maxiter=iters, tol=-1.0, beta=beta, lmd=lmd,
                                      closed_form_init=True)
        g = co.rbf_kernel(src, src, beta).astype(np.float64)
        want = src + g @ p["w"]
        got = res.transformation.transform(src)
        ext = float(np.max(np.abs(want)))
        e_t = float(np.max(np.abs(got - want))) / ext
        e_s = abs(res.sigma2 - s2) / max(abs(s2), 1e-300)
        err = max(e_t, 10.0 * e_s)
        worst = max(worst, err)
        flag = "" if err < 3e-4 else "   <-- OUT OF TOLERANCE"
        bad += bool(flag)
        print("case %2d m=%4d n=%4d dim=%d w=%.1f beta=%.1f lmd=%4.1f it=%d seed=%6d: T(Y) %.1e sigma2 %.1e%s" % (
            c, m, n, dim, w, beta, lmd, iters, seed, e_t, e_s, flag))
    print("%d cases, %d out of tolerance (T(Y) 3e-4 of the extent, sigma2 3e-5), worst %.2e, %.0f s" % (cases, bad, worst, time.time() - t0))


if __name__ == "__main__":
    main()
