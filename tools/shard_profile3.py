#!/usr/bin/env python
"""Dense- and late-regime E-step of one rank's shard (C1 / world) against the segment counts."""
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
states = {}
for it in range(30):
    if it in (4, 14, 29):
        states[it] = plan.get_params()
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
cy, cx = reg._cy, reg._cx
rows = dist.spatial_shard(tgt, 0, world) if world > 1 else np.arange(n)
for segs in ((0, 0),) if os.environ.get("ONLY_AUTO") else ((0, 0), (24, 6), (49, 12), (98, 24), (196, 48), (256, 48)):
    p2 = engine.CpdPlan()
    p2.set_source(src - cy)
    p2.set_target(tgt[rows] - cx, n_global=n)
    p2.set_tuning(0, segs[0], 0, segs[1])
    out = []
    for it, st in sorted(states.items()):
        p2.set_params(st)
        p2.estep(0.0)
        best = None
        for _ in range(3):
            ms = p2.estep_timed(0.0)
            if best is None or ms["total"] < best["total"]:
                best = ms
        out.append("it%2d col %.3f row %.3f tot %.3f" % (it, best["colpass"], best["rowpass"], best["total"]))
    print("segs %3d/%2d | %s" % (segs[0], segs[1], " | ".join(out)))
    p2.close()
