#!/usr/bin/env python
"""Randomised parity sweep for the sizes where the matrix-core engines - and with them the fused single sweep of rigid iterations -
run (both clouds >= 8192 points): random sizes 8k ... 60k on either side, 2-D / 3-D, outlier weights, update_scale on / off, a
random starting rotation, offsets far from the origin, 1 ... 24 iterations (through the hand-over from the fused sweep to the two
culled sweeps), rigid AND affine (which keeps the two sweeps), against the reference's loop with the C / OpenMP E-step.

    python tools/fuzz_fused.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpd_c, cpd_numpy as co  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst, bad, t0 = 0.0, 0, time.time()
    for c in range(cases):
        m = int(rng.integers(8192, 60000))
        n = int(rng.integers(8192, 60000))
        dim = int(rng.choice([2, 3, 3]))
        kind = str(rng.choice(["rigid", "rigid", "affine"]))
        w = float(rng.choice([0.0, 0.0, 0.1, 0.4]))
        iters = int(rng.integers(1, 25))
        upd = bool(rng.integers(0, 2)) if kind == "rigid" else True
        seed = int(rng.integers(0, 10 ** 6))
        if kind == "rigid":
            src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=seed)
        else:
            src, tgt, _ = synthetic.affine_pair(n, m=m, seed=seed)
        if dim == 2:
            src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
        if rng.random() < 0.3:
            off = rng.uniform(-300, 300, dim)
            src, tgt = src + off, tgt + off
        init = None
        if kind == "rigid" and rng.random() < 0.4:
            ang = np.deg2rad(rng.uniform(-25, 25))
            rot = np.identity(dim)
            rot[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
            scale0 = float(rng.uniform(0.9, 1.1))
            # a starting transform that turns the source about ITS OWN centroid (clouds may sit hundreds of units from the origin:
            # a rotation about the origin would throw the source far from the target - there every P underflows float32, n_p is
            # 1e-77 in the reference's float64 and nothing is left to compare)
            cen = src.mean(axis=0)
            init = dict(rot=rot, t=rng.uniform(-0.05, 0.05, dim) + cen - scale0 * rot @ cen, scale=scale0)
        kw = dict(update_scale=upd) if kind == "rigid" else {}
        if init is not None:
            kw["tf_init_params"] = init
        res = cpd.registration_cpd(src, tgt, kind, w=w, maxiter=iters, tol=-1.0, **kw)
        if kind == "rigid":
            p = dict(init) if init is not None else dict(rot=np.identity(dim), t=np.zeros(dim), scale=1.0)
        else:
            p = dict(b=np.identity(dim), t=np.zeros(dim))
        s2 = co.squared_kernel_sum_closed_form(src, tgt)
        for _ in range(iters):  # the reference's loop (cpd.py:110-113), fp64
            es = co.EstepResult(*cpd_c.expectation_step(co.transform(kind, p, src), tgt, s2, w))
            p, s2, _q = co.mstep_rigid(src, tgt, es, update_scale=upd) if kind == "rigid" else co.mstep_affine(src, tgt, es)
        lin, want = (res.transformation.rot, p["rot"]) if kind == "rigid" else (res.transformation.b, p["b"])
        e_lin = float(np.max(np.abs(lin - want))) / float(np.max(np.abs(want)))
        e_t = float(np.max(np.abs(res.transformation.t - p["t"]))) / max(1.0, float(np.max(np.abs(p["t"]))))
        e_s = abs(res.sigma2 - s2) / s2
        err = max(e_lin, e_t, 10.0 * e_s)
        worst = max(worst, err)
        flag = "" if err < 1e-4 else "   <-- OUT OF TOLERANCE"
        bad += bool(flag)
        print("case %2d m=%5d n=%5d dim=%d %-6s w=%.1f scale=%d init=%d it=%2d seed=%6d: lin %.1e t %.1e sigma2 %.1e%s" % (
            c, m, n, dim, kind, w, upd, init is not None, iters, seed, e_lin, e_t, e_s, flag), flush=True)
    print("%d cases, %d out of tolerance, worst %.2e (max of transform errors and 10 x sigma2 error; tolerance 1e-4), %.0f s" % (
        cases, bad, worst, time.time() - t0))


if __name__ == "__main__":
    main()
