#!/usr/bin/env python
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, synthetic
from probreg_amd.engine import CpdPlan
n = 100000
src, tgt, (r, t, _) = synthetic.rigid_pair(n, seed=0)
z = src @ r.T + t
s32, t32 = (z - tgt.mean(0)).astype(np.float32), (tgt - tgt.mean(0)).astype(np.float32)
plan = CpdPlan(); plan.set_source(s32); plan.set_target(t32)
p = np.zeros(32); p[[0,4,8,12]] = 1.0
for s2 in (1e-1, 1e-3, 1e-4, 3.3e-5, 1e-6, 1e-9):
    for seg in (0, 8, 64):
        plan.set_tuning(0, seg, 0, seg)
        p[13] = s2; plan.set_params(p)
        plan.estep(0.0); plan.estep(0.0)
        ms = min((plan.estep_timed(0.0) for _ in range(3)), key=lambda d: d["total"])
        print("sigma2 %.1e seg %2d: transform %.3f colpass %.3f colfinal %.3f rowpass %.3f moments %.3f" % (s2, seg, ms["transform"], ms["colpass"], ms["colfinal"], ms["rowpass"], ms["moments"]))
