#!/usr/bin/env python
"""What ONE rank of N does over the bench window (EM iterations 0..K-1 of C1), measured on one GPU: the unsharded registration
is run once and its parameter block saved before every iteration; rank 0's shard of the target (its run of the Morton order,
probreg_amd.dist.spatial_shard) is then put in each of those states and the E-step timed back to back (no host synchronisation
in between).  The M-step (one 1-thread kernel) and the 32-double all-reduce are not part of it.

    python tools/shard_window.py [n] [K]          -> a table per world size and the sums the 8-GPU projection of DESIGN.md uses
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import _lib, cpd, dist, engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
states = []
for it in range(K):
    states.append(plan.get_params())
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
cy, cx = reg._cy, reg._cx
table = {}
for world in (1, 2, 4, 8):
    rows = dist.spatial_shard(tgt, 0, world) if world > 1 else np.arange(n)
    p2 = engine.CpdPlan()
    p2.set_source(src - cy)
    p2.set_target(tgt[rows] - cx, n_global=n)
    p2.init_sums()  # (as registration does: the local target's sums decide where the lean row pass may run)
    ms = []
    for it, st in enumerate(states):
        # a faithful E-step needs the PREVIOUS iteration's column minima as seeds: run the previous state first
        p2.set_params(states[max(it - 1, 0)])
        p2.estep(0.0)
        p2.set_params(st)
        for _ in range(3):
            p2.estep(0.0)
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            p2.estep(0.0)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / reps * 1e3)
    table[world] = ms
    p2.close()
print("# E-step of rank 0 (ms), RigidCPD N=M=%d, EM iterations 0..%d, back to back" % (n, K - 1))
print("%3s %10s %10s %10s %10s" % ("it", "1 rank", "2 ranks", "4 ranks", "8 ranks"))
for it in range(K):
    print("%3d %10.3f %10.3f %10.3f %10.3f" % (it, table[1][it], table[2][it], table[4][it], table[8][it]))
tot = {w: sum(table[w]) for w in table}
print("sum %10.3f %10.3f %10.3f %10.3f" % (tot[1], tot[2], tot[4], tot[8]))
fixed = 0.012 * K  # k_mstep + k_reduce gaps measured on one GPU (~12 us per iteration)
for comm_us in (0.0, 30.0, 60.0):
    line = "with %2.0f us per iteration for the 32-double all-reduce:" % comm_us
    for w in (2, 4, 8):
        line += "  %d ranks %.2fx" % (w, (tot[1] + fixed) / (tot[w] + fixed + comm_us * 1e-3 * K))
    print(line)
