#!/usr/bin/env python
"""Time BCPD iterations (weighted E-step + Woodbury M-step solve + host algebra) at a given size.

    python tools/time_bcpd.py [M=N] [iterations]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import bcpd, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    src, tgt = synthetic.nonrigid_pair(n, seed=0)
    src, tgt = src * 10.0, tgt * 10.0  # object ~ 20 units across: the c = 1 kernel then has a sensible width
    reg = bcpd.CombinedBCPD(src)
    stamps = []
    reg.set_callbacks([lambda tr: (torch.cuda.synchronize(), stamps.append(time.perf_counter()))])
    t0 = time.perf_counter()
    res = reg._initialize(tgt)
    torch.cuda.synchronize()
    print("N=M=%d setup (upload + G build + sigma2 init) %.3f s" % (n, time.perf_counter() - t0))
    # one instrumented iteration
    plan = reg._plan
    t0 = time.perf_counter()
    trans = reg.registration(tgt, w=0.05, maxiter=iters, tol=-1.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    prev = t0
    for i, s in enumerate(stamps):
        print("iter %d: %.1f ms" % (i, (s - prev) * 1e3))
        prev = s
    print("total %.1f ms for %d iterations; scale=%.5f" % ((t1 - t0) * 1e3, iters, trans.rigid_trans.scale))
    # the solve alone
    nu = np.ones(n)
    resid = np.zeros((n, 3))
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.bcpd_solve(2.0, 10.0, resid, nu)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("bcpd_solve alone: %.1f ms  (%.1f TFLOP/s f64 on 4/3 M^3)" % (dt * 1e3, 4.0 / 3.0 * n ** 3 / dt / 1e12))
    ms = plan.estep_timed(0.05)
    print("E-step kernels (ms):", {k: round(v, 3) for k, v in ms.items()})


if __name__ == "__main__":
    main()
