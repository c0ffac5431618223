#!/usr/bin/env python
"""Whole EM iterations of rank 0 of `world` at the state before EM iteration `it` of C1, back to back - the target of a
rocprofv3 --kernel-trace run (tools/rocpd_summary.py: kernel durations and the idle gaps between consecutive kernels).
    rocprofv3 --kernel-trace -d out -o t -- python tools/shard_timeline.py 8 19 [reps]"""
import ctypes
import os
import sys

import numpy as np

os.environ.setdefault("PROBREG_NATIVE_RCCL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import _lib, cpd, dist, engine, synthetic  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
it_state = int(sys.argv[2]) if len(sys.argv) > 2 else 19
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
n = 100000
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
comm = plan._comm
prev = None
for it in range(it_state):
    prev = plan.get_params()
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
state = plan.get_params()
rows = dist.spatial_shard(tgt, 0, world) if world > 1 else np.arange(n)
p2 = engine.CpdPlan()
p2.set_source(src - reg._cy)
p2.set_target(tgt[rows] - reg._cx, n_global=n)
p2.init_sums()
p2.set_moments_only(1)
ptr = ctypes.c_void_p()
_lib.check(_lib.lib.prg_cpd_params_ptr(p2._h, ctypes.byref(ptr)))


class _V(object):
    __cuda_array_interface__ = {"shape": (_lib.PRG_NPARAMS,), "typestr": "<f8", "data": (ptr.value, False), "version": 2}


view = torch.as_tensor(_V(), device="cuda:%d" % p2.device)
saved = torch.from_numpy(state.copy()).cuda()
p2.set_params(prev if prev is not None else state)
p2.estep(0.0)
p2.set_params(state)
if comm is not None:
    p2.set_comm(comm)
for _ in range(reps):
    view.copy_(saved)
    p2.estep(0.0)
    p2.mstep(_lib.PRG_TF_RIGID, True)
torch.cuda.synchronize()
print("world %d, state before iteration %d: %d whole iterations enqueued" % (world, it_state, reps))
