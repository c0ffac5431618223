#!/usr/bin/env python
"""What EVERY rank of N does over the bench window (EM iterations 0..K-1 of C1), measured on one GPU: the unsharded registration
is run once and its parameter block saved before every iteration; each rank's shard of the target (its part of the spatial order,
probreg_amd.dist.spatial_shard) is then put in each of those states and timed back to back (no host synchronisation in between);
an N-rank iteration is priced at its SLOWEST rank (the per-iteration all-reduce is a barrier) and the projection is made from that:

  E-step        prg_cpd_estep alone (round 3's table)
  iteration     the whole EM iteration as a rank issues it: prg_cpd_estep ending with the library's own ncclAllReduce of the
                32-double moment block on the plan's stream (a ONE-rank RCCL communicator here: the launch and the kernel of
                the collective are in the measurement, the peers' latency is not), then k_mstep - with the state restored by
                a 256-byte device-to-device copy on the same stream, whose cost is measured on its own and subtracted

    python tools/shard_window.py [n] [K] [rigid|affine]   -> tables per world size and the sums the 8-GPU projection of DESIGN.md uses
                                                             (C1: 100000 20 rigid; C2, the configuration BASELINE.md quotes the
                                                             8-GPU target on: 200000 20 affine)
"""
import ctypes
import os
import sys
import time

import numpy as np

os.environ.setdefault("PROBREG_NATIVE_RCCL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import _lib, cpd, dist, engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
KIND = sys.argv[3] if len(sys.argv) > 3 else "rigid"
REPS = 30 if n <= 100000 else 10
if KIND == "rigid":
    src, tgt, _ = synthetic.rigid_pair(n, seed=0)
    reg, KIND_ID = cpd.RigidCPD(src), _lib.PRG_TF_RIGID
else:
    src, tgt, _ = synthetic.affine_pair(n, seed=0)
    reg, KIND_ID = cpd.AffineCPD(src), _lib.PRG_TF_AFFINE
reg._initialize(tgt)
plan = reg._plan
if KIND == "rigid":
    plan.set_moments_only(1)
comm = plan._comm
assert comm is not None, "library-side RCCL communicator unavailable"
states = []
for it in range(K):
    states.append(plan.get_params())
    plan.estep(0.0)
    plan.mstep(KIND_ID, True)
cy, cx = reg._cy, reg._cx


def params_view(p):
    ptr = ctypes.c_void_p()
    _lib.check(_lib.lib.prg_cpd_params_ptr(p._h, ctypes.byref(ptr)))

    class _V(object):
        __cuda_array_interface__ = {"shape": (_lib.PRG_NPARAMS,), "typestr": "<f8", "data": (ptr.value, False), "version": 2}

    return torch.as_tensor(_V(), device="cuda:%d" % p.device)


def timed(fn):
    """ms per call: the faster of two back-to-back batches of REPS calls (one batch now and then catches a hiccup of the box -
    5 ms on a 0.5 ms iteration in the first round-6 replay of C2 - which a max over ranks would carry into the projection)."""
    for _ in range(3):
        fn()
    best = None
    for _batch in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / REPS * 1e3
        best = dt if best is None else min(best, dt)
    return best


ALL_RANKS = os.environ.get("SHARD_WINDOW_ALL_RANKS", "1") != "0"   # 0: rank 0 only (rounds 3 - 5)


def replay(world, rank):
    """(E-step ms, whole-iteration ms, copy ms) per EM iteration of `rank`'s shard of `world`."""
    rows = dist.spatial_shard(tgt, rank, world) if world > 1 else np.arange(n)
    p2 = engine.CpdPlan()
    p2.set_source(src - cy)
    p2.set_target(tgt[rows] - cx, n_global=n)
    p2.init_sums()  # (as registration does: the local target's sums decide where the lean row pass may run)
    if KIND == "rigid":
        p2.set_moments_only(1)  # (as the registration's own loop: rigid iterations run single sweeps)
    view = params_view(p2)
    es, its, cps = [], [], []
    for it, st in enumerate(states):
        saved = torch.from_numpy(st.copy()).cuda()
        # a faithful E-step needs the PREVIOUS iteration's column minima as seeds: run the previous state first
        p2.set_comm(None)
        p2.set_params(states[max(it - 1, 0)])
        p2.estep(0.0)
        p2.set_params(st)
        es.append(timed(lambda: p2.estep(0.0)))
        p2.set_comm(comm)

        def iteration():
            view.copy_(saved)
            p2.estep(0.0)                       # ... + ncclAllReduce(moments) on the plan's stream
            p2.mstep(KIND_ID, True)

        its.append(timed(iteration))
        cps.append(timed(lambda: view.copy_(saved)))
    p2.set_comm(None)
    p2.close()
    return es, its, cps


if os.environ.get("SHARD_TRACE"):
    # SHARD_TRACE=world,rank,iteration[,reps]: that one whole iteration (E-step + all-reduce + M-step) in a loop and nothing else -
    # under `rocprofv3 --kernel-trace --stats` this is the per-kernel cost of one rank's iteration (tools/gpu_session.sh shard_trace)
    f = [int(v) for v in os.environ["SHARD_TRACE"].split(",")]
    world, rank, it = f[0], f[1], f[2]
    reps = f[3] if len(f) > 3 else 200
    rows = dist.spatial_shard(tgt, rank, world) if world > 1 else np.arange(n)
    p2 = engine.CpdPlan()
    p2.set_source(src - cy)
    p2.set_target(tgt[rows] - cx, n_global=n)
    p2.init_sums()
    if KIND == "rigid":
        p2.set_moments_only(1)
    view = params_view(p2)
    saved = torch.from_numpy(states[it].copy()).cuda()
    for j in range(max(it - 2, 0), it):   # the engine's memory and the column minima of the iterations before
        p2.set_params(states[j])
        p2.estep(0.0)
    p2.set_comm(comm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        view.copy_(saved)
        p2.estep(0.0)
        p2.mstep(KIND_ID, True)
    torch.cuda.synchronize()
    print("# rank %d of %d, EM iteration %d: %.4f ms per whole iteration (incl. the state-restoring copy), engine %s" % (
        rank, world, it, (time.perf_counter() - t0) / reps * 1e3, p2.last_estep_engines()))
    p2.set_comm(None)
    p2.close()
    sys.exit(0)

WORLDS = (1, 2, 4, 8)
per_rank = {}   # (world, rank) -> (E-step, whole iteration minus the copy) per EM iteration
copy_us = []
for world in WORLDS:
    for rank in range(world if ALL_RANKS else 1):
        if world > 1:
            REPS = 12 if n <= 100000 else 6
        es, its, cps = replay(world, rank)
        per_rank[(world, rank)] = (np.array(es), np.array(its) - np.array(cps))
        copy_us.append(1e3 * float(np.mean(cps)))

name = "RigidCPD" if KIND == "rigid" else "AffineCPD"
print("# %s N=M=%d, EM iterations 0..%d, every rank's shard of 1 / 2 / 4 / 8 replayed back to back on one MI355X (ms)" % (name, n, K - 1))
print("# an iteration of an N-rank run lasts as long as its SLOWEST rank (the all-reduce is a barrier): per iteration the MAX over the")
print("# ranks' whole iterations = E-step + ncclAllReduce(32 fp64, 1-rank communicator, plan's stream) + M-step, minus the state-restoring copy;")
print("# skew = max / mean over the ranks of that iteration (1.00: perfectly balanced shards)")
ranks_of = {w: [r for r in range(w) if (w, r) in per_rank] for w in WORLDS}
mx = {w: np.max([per_rank[(w, r)][1] for r in ranks_of[w]], axis=0) for w in WORLDS}
mean = {w: np.mean([per_rank[(w, r)][1] for r in ranks_of[w]], axis=0) for w in WORLDS}
emx = {w: np.max([per_rank[(w, r)][0] for r in ranks_of[w]], axis=0) for w in WORLDS}
print("%3s %9s %9s %9s %9s | %6s %6s %6s | %9s" % ("it", "1 rank", "2 ranks", "4 ranks", "8 ranks", "skew2", "skew4", "skew8", "8: rank 0"))
for it in range(K):
    print("%3d %9.3f %9.3f %9.3f %9.3f | %6.3f %6.3f %6.3f | %9.3f" % (
        (it,) + tuple(mx[w][it] for w in WORLDS) + tuple(mx[w][it] / mean[w][it] for w in (2, 4, 8)) + (per_rank[(8, 0)][1][it],)))
tf = {w: float(np.sum(mx[w])) for w in WORLDS}
te = {w: float(np.sum(emx[w])) for w in WORLDS}
print("sum %9.3f %9.3f %9.3f %9.3f | %6.3f %6.3f %6.3f | %9.3f" % (
    tuple(tf[w] for w in WORLDS) + tuple(tf[w] / float(np.sum(mean[w])) for w in (2, 4, 8)) + (float(np.sum(per_rank[(8, 0)][1])),)))
print("# per rank of 8, sum over the window (ms): " + "  ".join("%d: %.3f" % (r, float(np.sum(per_rank[(8, r)][1]))) for r in ranks_of[8]))
print("# state-restoring copy (subtracted): %.1f us mean; E-step alone, max over ranks, summed: %s" % (
    float(np.mean(copy_us)), "  ".join("%d ranks %.3f" % (w, te[w]) for w in WORLDS)))
print("# E-step only (max over ranks):          2 ranks %.2fx  4 ranks %.2fx  8 ranks %.2fx" % tuple(te[1] / te[w] for w in (2, 4, 8)))
for peer_us in (0.0, 10.0, 20.0, 30.0):
    line = "# whole iterations (max over ranks), + %2.0f us of peer latency per all-reduce (NOT measurable on one GPU):" % peer_us
    for w in (2, 4, 8):
        line += "  %d ranks %.2fx" % (w, tf[1] / (tf[w] + peer_us * 1e-3 * K))
    print(line)
if ALL_RANKS:
    r0 = {w: float(np.sum(per_rank[(w, 0)][1])) for w in WORLDS}
    print("# (rank 0 alone, as rounds 3 - 5 projected:  2 ranks %.2fx  4 ranks %.2fx  8 ranks %.2fx)" % tuple(r0[1] / r0[w] for w in (2, 4, 8)))
