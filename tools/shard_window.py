#!/usr/bin/env python
"""What ONE rank of N does over the bench window (EM iterations 0..K-1 of C1), measured on one GPU: the unsharded registration
is run once and its parameter block saved before every iteration; rank 0's shard of the target (its run of the Morton order,
probreg_amd.dist.spatial_shard) is then put in each of those states and timed back to back (no host synchronisation in between):

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
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3


estep_ms, iter_ms, copy_ms = {}, {}, {}
for world in (1, 2, 4, 8):
    rows = dist.spatial_shard(tgt, 0, world) if world > 1 else np.arange(n)
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
    estep_ms[world], iter_ms[world], copy_ms[world] = es, its, cps
    p2.set_comm(None)
    p2.close()

print("# rank 0 of N, %s N=M=%d, EM iterations 0..%d, back to back on one MI355X (ms)" % ("RigidCPD" if KIND == "rigid" else "AffineCPD", n, K - 1))
print("# E-step alone | whole iteration = E-step + ncclAllReduce(32 fp64, 1-rank communicator, plan's stream) + M-step, minus the state-restoring copy")
print("%3s %9s %9s %9s %9s | %9s %9s %9s %9s" % ("it", "1 rank", "2 ranks", "4 ranks", "8 ranks", "1 rank", "2 ranks", "4 ranks", "8 ranks"))
full = {w: [a - c for a, c in zip(iter_ms[w], copy_ms[w])] for w in iter_ms}
for it in range(K):
    print("%3d %9.3f %9.3f %9.3f %9.3f | %9.3f %9.3f %9.3f %9.3f" % ((it,) + tuple(estep_ms[w][it] for w in (1, 2, 4, 8))
                                                                      + tuple(full[w][it] for w in (1, 2, 4, 8))))
te = {w: sum(estep_ms[w]) for w in estep_ms}
tf = {w: sum(full[w]) for w in full}
print("sum %9.3f %9.3f %9.3f %9.3f | %9.3f %9.3f %9.3f %9.3f" % (tuple(te[w] for w in (1, 2, 4, 8)) + tuple(tf[w] for w in (1, 2, 4, 8))))
print("# state-restoring copy (subtracted): %.1f us mean; fixed work per iteration beyond the E-step (collective launch + kernel, M-step): "
      "1 rank %.1f us, 8 ranks %.1f us" % (1e3 * np.mean(copy_ms[8]), 1e3 * (tf[1] - te[1]) / K, 1e3 * (tf[8] - te[8]) / K))
print("# E-step only:          2 ranks %.2fx  4 ranks %.2fx  8 ranks %.2fx" % tuple(te[1] / te[w] for w in (2, 4, 8)))
for peer_us in (0.0, 10.0, 20.0, 30.0):
    line = "# whole iterations, + %2.0f us of peer latency per all-reduce (NOT measurable on one GPU):" % peer_us
    for w in (2, 4, 8):
        line += "  %d ranks %.2fx" % (w, tf[1] / (tf[w] + peer_us * 1e-3 * K))
    print(line)
