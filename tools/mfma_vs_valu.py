#!/usr/bin/env python
"""Matrix-core vs vector-pipe E-step along one registration: per EM iteration, from the SAME state, the time of both
engines' E-steps and how far their M-step results are apart (sigma2, rotation).  Decides the precision bound and the
speed crossover of prg_cpd_set_dense_engine.   usage: mfma_vs_valu.py [n] [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 22
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
plan.set_dense_engine(0)
plan.estep(0.0)
plan.mstep(_lib.PRG_TF_RIGID, True)
print("iter sigma2      nk      | valu: col row total ms | mfma: col row total ms | d sigma2 rel, d rot, d t")
for it in range(1, iters):
    state = plan.get_params()
    res = {}
    for eng in (0, 2):
        plan.set_dense_engine(eng)
        plan.set_params(state)
        ms = plan.estep_timed(0.0)
        used = plan.last_estep_engine()
        pc = plan.pair_counts()
        plan.mstep(_lib.PRG_TF_RIGID, True)
        res[eng] = (ms, plan.get_params(), used, pc)
    a, b = res[0][1], res[2][1]
    nk = 1.4426950408889634 / (2.0 * state[13])
    print("%3d  %.3e %8.1f | %.3f %.3f %.3f | %.3f %.3f %.3f (mfma %d) | %.2e %.2e %.2e" % (
        it, state[13], nk, res[0][0]["colpass"], res[0][0]["rowpass"], res[0][0]["total"], res[2][0]["colpass"],
        res[2][0]["rowpass"], res[2][0]["total"], res[2][2], abs(a[13] - b[13]) / a[13], np.max(np.abs(a[:9] - b[:9])),
        np.max(np.abs(a[9:12] - b[9:12]))), "| pairs/1e9 valu %.2f %.2f mfma %.2f %.2f" % (
        res[0][3][0] / 1e9, res[0][3][1] / 1e9, res[2][3][0] / 1e9, res[2][3][1] / 1e9))
    plan.set_dense_engine(0)   # advance along the vector-pipe trajectory
    plan.set_params(a)
