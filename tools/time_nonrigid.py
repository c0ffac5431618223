#!/usr/bin/env python
"""Time the non-rigid EM iteration (E-step + fp64 Cholesky M-step) at a given size."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    src, tgt = synthetic.nonrigid_pair(n, seed=0)
    t0 = time.perf_counter()
    reg = cpd.NonRigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    torch.cuda.synchronize()
    print("N=M=%d setup (upload + G build) %.3f s" % (n, time.perf_counter() - t0))
    for it in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms = plan.estep_timed(0.0)
        t1 = time.perf_counter()
        plan.mstep_nonrigid(2.0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        p = plan.get_params()
        flops = (n ** 3) / 3.0
        print("iter %d: estep %.2f ms (transform %.2f col %.2f row %.2f)  mstep %.1f ms  (chol %.1f TFLOP/s f64 equiv)  sigma2=%.6e"
              % (it, (t1 - t0) * 1e3, ms["transform"], ms["colpass"], ms["rowpass"], (t2 - t1) * 1e3,
                 flops / (t2 - t1) / 1e12, p[13]))


if __name__ == "__main__":
    main()
