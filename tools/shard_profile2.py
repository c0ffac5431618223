#!/usr/bin/env python
"""Late-regime E-step floor of one rank's shard (C1 / 8): segment sweep and the skip-everything floor."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, dist, engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
for it in range(30):
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
st = plan.get_params()
cy, cx = reg._cy, reg._cx
rows = dist.spatial_shard(tgt, 0, world) if world > 1 else np.arange(n)
for segs in ((0, 0), (16, 16), (32, 32), (64, 64), (128, 64), (256, 64)):
    p2 = engine.CpdPlan()
    p2.set_source(src - cy)
    p2.set_target(tgt[rows] - cx, n_global=n)
    p2.set_tuning(0, segs[0], 0, segs[1])
    for label, s2 in (("late", st[13]), ("floor", 1e-9)):
        q = st.copy()
        q[13] = s2
        p2.set_params(q)
        p2.estep(0.0)
        best = None
        for _ in range(3):
            ms = p2.estep_timed(0.0)
            if best is None or ms["total"] < best["total"]:
                best = ms
        print("segs col/row %2d/%2d %-5s sigma2 %.1e: transform %.3f colpass %.3f colfinal %.3f rowpass %.3f moments %.3f total %.3f"
              % (segs[0], segs[1], label, s2, best["transform"], best["colpass"], best["colfinal"], best["rowpass"],
                 best["moments"], best["total"]))
    p2.close()
