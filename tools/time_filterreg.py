#!/usr/bin/env python
"""Time FilterReg EM iterations (C4: N = M = 500k, 5 % outliers) on the GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import filterreg, math_utils as mu, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    src, tgt, _ = synthetic.filterreg_pair(n, seed=0)
    reg = filterreg.RigidFilterReg(src, update_sigma2=True)
    sigma2 = max(mu.squared_kernel_sum(src, tgt), 1e-4)
    plan = reg._ensure_plan(tgt)
    rot, t = np.identity(3), np.zeros(3)
    torch.cuda.synchronize()
    plan.set_state(rot, t, sigma2)
    for it in range(iters):
        t0 = time.perf_counter()
        size, blur = plan.estep()
        t1 = time.perf_counter()
        out = plan.mstep(0.05, True, "pt2pt", 1e-4)
        t2 = time.perf_counter()
        print("iter %2d: estep %.3f ms (lattice %d vertices, blur=%d)  mstep %.3f ms  sigma2=%.5e q=%.5e"
              % (it, (t1 - t0) * 1e3, size, blur, (t2 - t1) * 1e3, out[15], out[13]))


if __name__ == "__main__":
    main()
