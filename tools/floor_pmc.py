#!/usr/bin/env python
"""E-steps at a state where (almost) every block is culled - target of rocprofv3 --pmc runs on the sweep floor."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
s2 = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-9
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
for it in range(30):
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
st = plan.get_params()
st[13] = s2
plan.set_params(st)
for _ in range(8):
    ms = plan.estep_timed(0.0)
print(ms)
