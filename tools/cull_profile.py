#!/usr/bin/env python
"""Per-iteration E-step kernel times for C1 with the culled sweeps (how much of the ideal saving is realised)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src); reg._initialize(tgt); plan = reg._plan
for it in range(30):
    s2 = plan.get_params()[13]
    ms = plan.estep_timed(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
    print("iter %2d sigma2 %.3e  transform %.3f colpass %.3f colfinal %.3f rowpass %.3f moments %.3f total %.3f" % (
        it, s2, ms["transform"], ms["colpass"], ms["colfinal"], ms["rowpass"], ms["moments"], ms["total"]))
