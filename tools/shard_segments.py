import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from probreg_amd import _lib, cpd, dist, engine, synthetic
n = 100000
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src); reg._initialize(tgt); plan = reg._plan
states = {}
for it in range(30):
    if it in (14, 19, 29): states[it] = plan.get_params()
    plan.estep(0.0); plan.mstep(_lib.PRG_TF_RIGID, True)
cy, cx = reg._cy, reg._cx
for world in (4, 8):
    rows = dist.spatial_shard(tgt, 0, world)
    for tune in ((0, 0, 0, 0), (0, 391, 0, 0), (0, 0, 0, int(np.ceil(len(rows) / 256))), (0, 391, 0, int(np.ceil(len(rows) / 256))), (0, 782, 0, int(np.ceil(len(rows) / 256)))):
        p2 = engine.CpdPlan(); p2.set_source(src - cy); p2.set_target(tgt[rows] - cx, n_global=n)
        p2.set_tuning(*tune)
        out = []
        for it, st in sorted(states.items()):
            p2.set_params(st)
            for _ in range(5): p2.estep(0.0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): p2.estep(0.0)
            torch.cuda.synchronize(); out.append("it%2d %.3f" % (it, (time.perf_counter() - t0) * 10))
        print("world %d tuning %s: %s" % (world, tune, " | ".join(out)))
        p2.close()
