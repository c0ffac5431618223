#!/usr/bin/env python
"""Matrix-core vs vector-pipe E-step along one registration: per EM iteration, from the SAME state, the time of both
engines' E-steps and how far their M-step results are apart (sigma2, rotation).  Decides the precision bound and the
speed crossover of prg_cpd_set_dense_engine.   usage: mfma_vs_valu.py [n] [iterations] [surface|volume|aniso] [world]

Clouds: the tube-like surface of every other workload, or two NON-surface ones - a uniform sample of the unit cube
(volume) and of a 10 : 1 : 1 box (aniso); there the target is the SAME sample, rotated, shifted, with noise (two
independent volume samples have no structure to lock on to: sigma2 would stay large for hundreds of iterations).  The last
columns say what the library's own switch (engine 1) chose from the same state and what that choice costs next to the
faster engine of each pass."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, cpd, dist, engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 22
kind = sys.argv[3] if len(sys.argv) > 3 else "surface"
if kind == "surface":
    src, tgt, _ = synthetic.rigid_pair(n, seed=0)
else:
    rng = np.random.default_rng(0)
    box = np.array([1.0, 1.0, 1.0]) if kind == "volume" else np.array([10.0, 1.0, 1.0])
    src = rng.random((n, 3)) * box
    rot = synthetic.rot_zx(12.0, 5.0)
    tgt = (src @ rot.T + np.array([0.05, -0.03, 0.02]) + 0.004 * rng.standard_normal((n, 3)))[rng.permutation(n)]
world = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # > 1: rank 0's shard of the target, states of the 1-rank trajectory
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
plan.set_dense_engine(0)
states = []
for it in range(iters):  # the trajectory (vector-pipe sweeps)
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
    states.append(plan.get_params())
n_local = tgt.shape[0]
if world > 1:
    rows = dist.spatial_shard(tgt, 0, world)
    plan = engine.CpdPlan()
    plan.set_source(src - reg._cy)
    plan.set_target(tgt[rows] - reg._cx, n_global=tgt.shape[0])
    plan.init_sums()  # (as registration does: the local target's sums decide where the lean row pass may run)
    n_local = len(rows)
print("%s cloud, M = %d, N = %d (rank 0 of %d: %d targets)" % (kind, src.shape[0], tgt.shape[0], world, n_local))
print("iter sigma2      nk      | valu: col row total ms | mfma: col row total ms | d sigma2 rel, d rot, d t | pairs | "
      "mfma sweeps: evaluated fraction col row, pairs per owned point col row | own switch: col row engine, col row ms, "
      "loss vs the faster engine col row")
mn = float(src.shape[0]) * float(n_local)
worst = [0.0, 0.0]
for it, state in enumerate(states[:-1], 1):
    res = {}
    for eng in (0, 2, 1):
        plan.set_dense_engine(eng)
        plan.set_params(state)
        # steady state of every mode: the switch goes by what the previous E-step's sweeps counted (the first E-step after
        # a mode change has nothing and takes the matrix cores, the second decides), and the work queue of the
        # vector-pipe sweeps sizes its units from the previous build
        for _ in range({0: 1, 2: 0, 1: 3}[eng]):
            plan.estep(0.0)
        ms = plan.estep_timed(0.0)
        used = plan.last_estep_engines()
        pc = plan.pair_counts()
        plan.mstep(_lib.PRG_TF_RIGID, True)
        res[eng] = (ms, plan.get_params(), used, pc)
    a, b = res[0][1], res[2][1]
    nk = 1.4426950408889634 / (2.0 * state[13])
    loss = [res[1][0][k] / min(res[0][0][k], res[2][0][k]) - 1.0 for k in ("colpass", "rowpass")]
    worst = [max(worst[0], loss[0]), max(worst[1], loss[1])]
    print("%3d  %.3e %8.1f | %.3f %.3f %.3f | %.3f %.3f %.3f (mfma %d %d) | %.2e %.2e %.2e" % (
        it, state[13], nk, res[0][0]["colpass"], res[0][0]["rowpass"], res[0][0]["total"], res[2][0]["colpass"],
        res[2][0]["rowpass"], res[2][0]["total"], res[2][2][0], res[2][2][1], abs(a[13] - b[13]) / a[13],
        np.max(np.abs(a[:9] - b[:9])), np.max(np.abs(a[9:12] - b[9:12]))), "| pairs/1e9 valu %.2f %.2f mfma %.2f %.2f" % (
        res[0][3][0] / 1e9, res[0][3][1] / 1e9, res[2][3][0] / 1e9, res[2][3][1] / 1e9),
        "| f %.4f %.4f r %6.0f %6.0f | own %d %d  %.3f %.3f  %+.0f%% %+.0f%%" % (
        res[2][3][0] / mn, res[2][3][1] / mn, res[2][3][0] / n_local, res[2][3][1] / src.shape[0], res[1][2][0], res[1][2][1],
        res[1][0]["colpass"], res[1][0]["rowpass"], 100 * loss[0], 100 * loss[1]))
print("largest loss of the library's own switch next to the faster engine: column pass %+.0f%%, row pass %+.0f%%" % (
    100 * worst[0], 100 * worst[1]))
