#!/usr/bin/env python
"""The two SINGLE sweeps of a rigid iteration from identical states along one registration: the fused matrix-core sweep
(prg_cpd_set_dense_engine(2): forced) against the owner sweep on the vector pipe (0), and what the library's own hand-over (1,
the default) chose and lost next to the faster of the two.  What `tools/mfma_vs_valu.py` is for the two-sweep engines.

    python tools/single_sweep_ab.py [n] [iterations] [surface|aniso|volume|clusters] [world] [rank]

world > 1: that rank's shard (probreg_amd.dist.spatial_shard) of the target, in the states of the one-rank trajectory.
Times are whole E-steps (transform + decision + sweep + column merge + final reduction), HIP events on the plan's stream."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzz_fused import make_clouds  # noqa: E402
from probreg_amd import _lib, cpd, dist, engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 24
shape = sys.argv[3] if len(sys.argv) > 3 else "surface"
world = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rank = int(sys.argv[5]) if len(sys.argv) > 5 else 0
src, tgt = make_clouds(np.random.default_rng(0), shape, "rigid", n, n, 0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
plan.set_moments_only(1)
states = []
for it in range(iters):  # the trajectory, as the registration's own loop runs it
    states.append(plan.get_params())
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
n_local = tgt.shape[0]
if world > 1:
    rows = dist.spatial_shard(tgt, rank, world)
    plan = engine.CpdPlan()
    plan.set_source(src - reg._cy)
    plan.set_target(tgt[rows] - reg._cx, n_global=tgt.shape[0])
    plan.init_sums()
    plan.set_moments_only(1)
    n_local = len(rows)
print("# %s cloud, M = N = %d (rank %d of %d: %d targets); whole single-sweep E-steps from the same state, ms" % (shape, n, rank, world, n_local))
print("# it   sigma2     | owner (vector pipe)   pairs/1e9 | matrix cores (forced)  pairs/1e9  single | library's own: engine ms   behind the faster")
worst, worst_it, total = 0.0, -1, {0: 0.0, 2: 0.0, 1: 0.0}
for it in range(1, len(states)):
    res = {}
    for eng in (0, 2, 1):
        plan.set_dense_engine(eng)
        for warm in ((it - 1, it) if eng != 1 else (max(it - 3, 0), max(it - 2, 0), it - 1, it)):
            plan.set_params(states[warm])   # (seeds, the switch's memory and the grid choice come from the iterations before)
            plan.estep(0.0)
        plan.set_params(states[it])
        ms = plan.estep_timed(0.0)["total"]
        pc, _pr = plan.pair_counts()
        res[eng] = (ms, pc, plan.last_estep_engine(), plan.last_estep_fused())
    best = min(res[0][0], res[2][0])
    loss = res[1][0] / best - 1.0
    for e in total:
        total[e] += res[e][0]
    if loss > worst:
        worst, worst_it = loss, it
    print("%4d  %.3e |        %7.3f         %8.3f |        %7.3f         %8.3f     %d   |       %d    %7.3f        %+5.1f %%" % (
        it, states[it][13], res[0][0], res[0][1] / 1e9, res[2][0], res[2][1] / 1e9, res[2][3], res[1][2], res[1][0], 100.0 * loss))
print("# sums over iterations 1..%d: owner only %.3f ms, matrix cores only %.3f ms, the library's hand-over %.3f ms; largest single-iteration "
      "loss of the hand-over next to the faster engine: %+.1f %% (iteration %d)" % (len(states) - 1, total[0], total[2], total[1], 100.0 * worst, worst_it))
