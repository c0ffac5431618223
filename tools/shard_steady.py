#!/usr/bin/env python
"""Back-to-back E-steps of one rank's shard (no host sync in between): steady-state time per E-step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import _lib, cpd, dist, engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
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
for world in (1, 2, 4, 8):
    rows = dist.spatial_shard(tgt, 0, world) if world > 1 else np.arange(n)
    p2 = engine.CpdPlan()
    p2.set_source(src - cy)
    p2.set_target(tgt[rows] - cx, n_global=n)
    p2.init_sums()  # (as registration does: the local target's sums decide where the lean row pass may run)
    out = []
    for it, st in sorted(states.items()):
        p2.set_params(st)
        for _ in range(5):
            p2.estep(0.0)
        torch.cuda.synchronize()
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps):
            p2.estep(0.0)
        torch.cuda.synchronize()
        out.append("it%2d %.3f ms" % (it, (time.perf_counter() - t0) / reps * 1e3))
        if world == 8 or world == 1:  # where the time of one E-step goes (HIP events between the kernels)
            ms = p2.estep_timed(0.0)
            out[-1] += " [" + " ".join("%s %.0f" % (k[:5], 1e3 * v) for k, v in ms.items()) + " us; mfma col %d]" % p2.last_estep_engine()
    print("world %d: back-to-back E-step of rank 0 | %s" % (world, " | ".join(out)))
    p2.close()
