#!/usr/bin/env python
"""The fused single sweep of a rigid iteration pushed beyond where the library uses it: both engines pinned to
the matrix cores, the fused sweep allowed at every amplification (prg_cpd_set_lean_factor(1e30)), along a C1-style registration.
Per EM iteration: sigma2's amplification, the E-step's time (HIP events), sigma2 after the M-step against the fp64 oracle from
the same state - how far does it stay accurate, and how far does it stay ahead of the two-sweep engines (compare the E-step
column with profiles/r4_c1_pairs_per_iteration.log)?        usage: fused_error.py [n] [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpd_c, cpd_numpy as co  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
mean_x2 = float(np.mean(np.sum((tgt - tgt.mean(0)) ** 2, axis=1)))
rows = {}
for mode in ("fused", "default"):
    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    if mode == "fused":
        plan.set_dense_engine(2)
        plan.set_lean_factor(1e30)
        plan.set_moments_only(1)
    else:
        plan.set_moments_only(2)
    for it in range(iters):
        st = reg._result_from_params(plan.get_params())
        ms = plan.estep_timed(0.0)
        fused = plan.last_estep_fused()
        reg._device_mstep(plan)
        out = reg._result_from_params(plan.get_params())
        err = float("nan")
        if mode == "fused" and (it % 2 == 1 or it >= 10):
            tr = st.transformation
            es = co.EstepResult(*cpd_c.expectation_step(co.transform("rigid", dict(rot=tr.rot, t=tr.t, scale=float(tr.scale)), src), tgt,
                                                        st.sigma2, 0.0))
            _, s2, _ = co.mstep_rigid(src, tgt, es)
            err = abs(out.sigma2 - s2) / s2
        rows.setdefault(it, {})[mode] = (st.sigma2, mean_x2 / (3.0 * st.sigma2), fused, ms["total"], err)
print("rigid n = %d: fused single sweep forced everywhere | the library's two-sweep default (prg_cpd_set_moments_only(2))" % n)
print(" it   sigma2 in   amplification  fused  E-step ms   sigma2 error vs oracle | default E-step ms")
for it in range(iters):
    f, d = rows[it]["fused"], rows[it]["default"]
    print("%3d  %.4e  %10.1f      %d    %8.3f    %10.2e            | %8.3f" % (it, f[0], f[1], f[2], f[3], f[4], d[3]))
