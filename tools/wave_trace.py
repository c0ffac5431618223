#!/usr/bin/env python
"""Per-wave timeline of the culled sweeps (needs the instrumented build: make -C probreg_amd/csrc trace).

    PROBREG_HIP_LIB=tools/bin/libprobreg_hip_trace.so python tools/wave_trace.py [n] [iterations before the probe]

Every wave records start / end shader clock, hardware id and the number of groups it evaluated; this prints how
the waves' lifetimes and start times are distributed - i.e. whether a sparse E-step is bound by launch rate, by a
few long waves, or by the sum of the work.
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from probreg_amd import _lib, cpd, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 30
src, tgt, _ = synthetic.rigid_pair(n, seed=0)
reg = cpd.RigidCPD(src)
reg._initialize(tgt)
plan = reg._plan
for it in range(warm):
    plan.estep(0.0)
    plan.mstep(_lib.PRG_TF_RIGID, True)
max_waves = 1 << 17
buf = torch.zeros(2 * max_waves * 4, dtype=torch.int64, device="cuda")
fn = _lib.lib.prg_debug_set_wave_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
fn.restype = ctypes.c_int
assert fn(ctypes.c_void_p(buf.data_ptr()), max_waves) == 0
ms = plan.estep_timed(0.0)
torch.cuda.synchronize()
print("sigma2 %.3e  kernel ms:" % plan.get_params()[13], {k: round(v, 3) for k, v in ms.items()})
tr = buf.cpu().numpy().reshape(2, max_waves, 4)
for k, name in ((0, "colpass"), (1, "rowpass")):
    t = tr[k]
    live = t[:, 1] > 0
    t = t[live]
    t0, t1, hw, groups = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    start = (t0 - t0.min()).astype(np.float64)
    dur = (t1 - t0).astype(np.float64)
    span = float(t1.max() - t0.min())
    print("%s: %d waves, span %.0f cycles; wave-cycles %.3g (avg %.0f, median %.0f, p99 %.0f, max %.0f)" % (
        name, len(t), span, dur.sum(), dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max()))
    print("   groups evaluated: total %d, waves with work %d (%.1f%%), max per wave %d" % (
        groups.sum(), (groups > 0).sum(), 100.0 * (groups > 0).mean(), groups.max()))
    for q in (10, 50, 90, 99, 100):
        print("   %3d%% of the waves had started by %.0f cycles" % (q, np.percentile(start, q)))
    busy = groups > 0
    if busy.any():
        print("   busy waves: mean %.0f cycles, %.0f cycles per group; idle waves: mean %.0f cycles" % (
            dur[busy].mean(), dur[busy].sum() / groups[busy].sum(), dur[~busy].mean() if (~busy).any() else 0))
    order = np.argsort(-dur)[:5]
    print("   longest waves (cycles, groups, start):", [(int(dur[i]), int(groups[i]), int(start[i])) for i in order])
