#!/usr/bin/env python
"""How far is sigma2 after an M-step from the fp64 oracle's when the matrix-core row pass runs without its residual sums
(k_rowpass_mfma<LEAN>)?  Along a C1-style registration, from identical states; run with
PRG_LEAN_FACTOR=1e30 to force the lean pass wherever the matrix-core row pass runs and with PRG_LEAN_FACTOR=0 for the full
one.   usage: lean_error.py [n] [kind: rigid|affine]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpd_c, cpd_numpy as co  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
kind = sys.argv[2] if len(sys.argv) > 2 else "rigid"
if kind == "rigid":
    src, tgt, _ = synthetic.rigid_pair(n, seed=0)
    reg = cpd.RigidCPD(src)
else:
    src, tgt, _ = synthetic.affine_pair(n, seed=0)
    reg = cpd.AffineCPD(src)
reg._initialize(tgt)
plan = reg._plan
plan.set_dense_engine(2)  # both sweeps on the matrix cores all the way
mean_x2 = float(np.mean(np.sum((tgt - tgt.mean(0)) ** 2, axis=1)))
print("PRG_LEAN_FACTOR=%s  %s n=%d  mean |x|^2 = %.4f" % (os.environ.get("PRG_LEAN_FACTOR", "(default)"), kind, n, mean_x2))
print(" it   sigma2 in    amplification   lean   sigma2 out (gpu)   rel. error vs oracle")
for it in range(15):
    st = reg._result_from_params(plan.get_params())
    plan.estep(0.0)
    lean = plan.last_estep_lean()
    reg._device_mstep(plan)
    out = reg._result_from_params(plan.get_params())
    if it in (1, 3, 5, 7, 9, 11, 12, 13, 14):
        tr = st.transformation
        p = dict(rot=tr.rot, t=tr.t, scale=float(tr.scale)) if kind == "rigid" else dict(b=tr.b, t=tr.t)
        es = co.EstepResult(*cpd_c.expectation_step(co.transform(kind, p, src), tgt, st.sigma2, 0.0))
        _, s2, _ = (co.mstep_rigid if kind == "rigid" else co.mstep_affine)(src, tgt, es)
        print("%3d  %.4e  %10.1f      %d     %.8e   %.2e" % (it, st.sigma2, mean_x2 / (3.0 * st.sigma2), lean, out.sigma2,
                                                              abs(out.sigma2 - s2) / s2))
