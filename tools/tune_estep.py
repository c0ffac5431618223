#!/usr/bin/env python
"""Sweep the E-step tuning knobs on the GPU and print per-kernel HIP-event timings (ms)."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probreg_amd import _lib, synthetic  # noqa: E402
from probreg_amd.engine import CpdPlan  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else n
    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=0)
    plan = CpdPlan()
    plan.set_source((src - src.mean(0)).astype(np.float32))
    plan.set_target((tgt - tgt.mean(0)).astype(np.float32))
    plan.init_sums()
    plan.init_params(None)
    for _ in range(3):
        plan.estep(0.0)
        plan.mstep(_lib.PRG_TF_RIGID, True)
    base = plan.get_params()
    print("N=%d M=%d sigma2=%.4e" % (n, m, base[13]))
    print("%6s %6s | %8s %8s %8s %8s %8s %8s" % ("R", "S", "transf", "colpass", "colfin", "rowpass", "moments", "total"))
    for r in (2, 4, -2, -4):
        for s in (0, 8, 16, 32, 64, 128):
            plan.set_tuning(r, s, r, s)
            plan.set_params(base)
            best = None
            for _ in range(3):
                ms = plan.estep_timed(0.0)
                if best is None or ms["total"] < best["total"]:
                    best = ms
            print("%6d %6d | %8.3f %8.3f %8.3f %8.3f %8.3f %8.3f" % (r, s, best["transform"], best["colpass"],
                  best["colfinal"], best["rowpass"], best["moments"], best["total"]))
    plan.close()


if __name__ == "__main__":
    main()
