#!/usr/bin/env python
"""Two identical FilterReg registrations must agree bit for bit (ordered splat), and so must two E-steps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import filterreg, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
src, tgt, _ = synthetic.filterreg_pair(n, seed=0)
out = []
for run in range(3):
    res = filterreg.registration_filterreg(src, tgt, sigma2=None, update_sigma2=True, w=0.05, maxiter=20, tol=-1.0)
    out.append((res.transformation.rot.copy(), res.transformation.t.copy(), res.sigma2, res.q))
    print("run %d: sigma2 %.17g q %.17g rot[0,0] %.17g" % (run, res.sigma2, res.q, res.transformation.rot[0, 0]))
same = all(np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1]) and out[0][2] == o[2] and out[0][3] == o[3]
           for o in out[1:])
print("bit-identical over 3 runs:", same)
