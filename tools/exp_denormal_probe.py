#!/usr/bin/env python
"""Does the raw v_exp_f32 (``__builtin_amdgcn_exp2f``) return denormals or flush them to zero?

Probe through the direct Gauss transform: one source at the origin with weight 1, targets at distances chosen so
that the exponent is -120 ... -152 in base 2 (h = 1).  Prints the value the kernel returns next to the exact one.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import gauss_transform as gt  # noqa: E402

LOG2E = 1.4426950408889634
args = np.array([-100.0, -120.0, -125.0, -126.0, -127.0, -130.0, -140.0, -148.0, -149.0, -150.0, -152.0])
d = np.sqrt(-args / LOG2E)
src = np.zeros((1, 3))
tgt = np.zeros((len(d), 3))
tgt[:, 0] = d
out = gt.GaussTransform(src, 1.0).compute(tgt, np.ones(1))
for a, o in zip(args, np.ravel(out)):
    print("exp2(%7.1f): kernel %.6e   exact %.6e" % (a, o, 2.0 ** a))
