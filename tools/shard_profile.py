#!/usr/bin/env python
"""What one of 8 ranks does per E-step at C1, for a shard in the caller's order vs a spatially compact shard.

Runs the unsharded registration for K iterations to get the EM state, then times the E-step of rank 0's shard
(source = all 100k points, target = 12.5k points) at that state with both shardings, on one GPU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, dist, engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
world = 8
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
states = {}
for it in range(30):
    if it in (3, 12, 16, 22, 29):
        states[it] = plan.get_params()
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
cy, cx = reg._cy, reg._cx
for label, rows in (("caller-order shard", np.arange(*dist.shard_bounds(n, 0, world))),
                    ("Morton-order shard", dist.spatial_shard(tgt, 0, world))):
    p2 = engine.CpdPlan()
    p2.set_source(src - cy)
    p2.set_target(tgt[rows] - cx, n_global=n)
    for it, st in sorted(states.items()):
        p2.set_params(st)
        p2.estep(0.0)          # first E-step at this state: builds the column-minimum seeds
        ms = p2.estep_timed(0.0)
        print("%-20s state of iteration %2d (sigma2 %.2e): colpass %.3f rowpass %.3f total %.3f ms" % (
            label, it, st[13], ms["colpass"], ms["rowpass"], ms["total"]))
    p2.close()
