#!/usr/bin/env python
"""Print sigma2 per EM iteration for C1 and the fraction of (m, n) pairs inside the exact-zero cut-off."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src); reg._initialize(tgt); plan = reg._plan
rng = np.random.default_rng(0)
si, ti = rng.choice(n, 2000, replace=False), rng.choice(n, 2000, replace=False)
for it in range(30):
    p = plan.get_params()
    s2 = p[13]
    res = reg._result_from_params(p)
    z = res.transformation.transform(src[si])
    d2 = ((z[:, None, :] - tgt[ti][None, :, :]) ** 2).sum(-1)
    rc2 = 150.0 * 2.0 * s2 / 1.4427
    print("iter %2d sigma2 %.4e cutoff radius %.3f frac pairs inside %.4f" % (it, s2, np.sqrt(rc2), float((d2 < rc2).mean())))
    plan.estep(0.0); plan.mstep(_lib.PRG_TF_RIGID, True)
