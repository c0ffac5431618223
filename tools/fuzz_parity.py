#!/usr/bin/env python
"""Randomised parity sweep: GPU registrations against the numpy oracle over random sizes, dimensions, outlier
weights and iteration counts (catches layout / tiling corner cases no hand-written test hits).

    python tools/fuzz_parity.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpd_numpy as co  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(float(np.max(np.abs(b))), 1e-300)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    bad = 0
    t0 = time.time()
    for c in range(cases):
        m = int(rng.choice([rng.integers(2, 40), rng.integers(40, 600), rng.integers(600, 4000)]))
        n = int(rng.choice([rng.integers(2, 40), rng.integers(40, 600), rng.integers(600, 4000)]))
        dim = int(rng.choice([2, 3]))
        kind = str(rng.choice(["rigid", "affine"]))
        w = float(rng.choice([0.0, 0.0, 0.1, 0.5, 0.9]))
        iters = int(rng.integers(1, 45))
        seed = int(rng.integers(0, 10 ** 6))
        if kind == "rigid":
            src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=seed)
        else:
            src, tgt, _ = synthetic.affine_pair(n, m=m, seed=seed)
        if dim == 2:
            src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
        if rng.random() < 0.3:  # offset far from the origin
            off = rng.uniform(-500, 500, dim)
            src, tgt = src + off, tgt + off
        if min(m, n) < dim + 2:
            # fewer points than the transformation has degrees of freedom: the cross-covariance is rank deficient and
            # the optimum not unique (numpy's SVD and the device Jacobi SVD complete it differently) - only the error
            # behaviour is comparable
            outcome = []
            for fn in (lambda: cpd.registration_cpd(src, tgt, kind, w=w, maxiter=iters, tol=-1.0),
                       lambda: co.registration(kind, src, tgt, w=w, maxiter=iters, tol=-1.0, closed_form_init=True)):
                try:
                    fn()
                    outcome.append("ok")
                except np.linalg.LinAlgError:
                    outcome.append("LinAlgError")
            print("case %2d m=%4d n=%4d dim=%d %-6s degenerate: gpu %s / oracle %s" % (c, m, n, dim, kind, outcome[0], outcome[1]))
            continue
        try:
            res = cpd.registration_cpd(src, tgt, kind, w=w, maxiter=iters, tol=-1.0)
            p, s2, q, _ = co.registration(kind, src, tgt, w=w, maxiter=iters, tol=-1.0, closed_form_init=True)
        except Exception as e:  # noqa: BLE001
            print("case %d m=%d n=%d dim=%d %s w=%.1f it=%d seed=%d: EXCEPTION %r" % (c, m, n, dim, kind, w, iters, seed, e))
            bad += 1
            continue
        lin = res.transformation.rot if kind == "rigid" else res.transformation.b
        e_lin = rel(lin, p["rot"] if kind == "rigid" else p["b"])
        e_t = float(np.max(np.abs(res.transformation.t - p["t"]))) / max(1.0, float(np.max(np.abs(p["t"]))))
        e_s = abs(res.sigma2 - s2) / max(abs(s2), 1e-300)
        err = max(e_lin, e_t, e_s * 10.0)  # sigma2 is held to 1e-5, the transform to 1e-4
        worst = max(worst, err)
        flag = "" if err < 1e-4 else "   <-- OUT OF TOLERANCE"
        if flag:
            bad += 1
        print("case %2d m=%4d n=%4d dim=%d %-6s w=%.1f it=%2d seed=%6d: lin %.1e t %.1e sigma2 %.1e%s" % (
            c, m, n, dim, kind, w, iters, seed, e_lin, e_t, e_s, flag))
    print("%d cases, %d out of tolerance, worst %.2e, %.0f s" % (cases, bad, worst, time.time() - t0))


if __name__ == "__main__":
    main()
