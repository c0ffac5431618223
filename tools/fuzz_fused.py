#!/usr/bin/env python
"""Randomised parity sweep for the sizes where the matrix-core engines - and with them the single sweeps of rigid iterations -
run (both clouds >= 8192 points), against the reference's loop with the C / OpenMP E-step (fp64).

Per case: random sizes on either side, 2-D / 3-D, outlier weight, update_scale on / off, a random starting rotation, 1 ... 32
EM iterations (through the fused matrix-core sweep, the hand-over, and the residual-form vector sweep / the two culled sweeps),
rigid AND affine (which keeps the two sweeps: lean matrix-core row pass, then the vector pipe), and a cloud SHAPE:

    surface     the tube-like surface of every workload (synthetic.rigid_pair / affine_pair)
    aniso       uniform sample of a 10 : 1 : 1 box, the target the same sample moved + noise
    volume      uniform sample of the unit cube, likewise
    clusters    two Gaussian blobs of different width, 6 units apart (empty space between: many culled blocks early on)
    far         any of the above 100 ... 1500 units from the origin

[r5] `big` cases append single-iteration comparisons at 100k ... 250k points (the oracle costs ~8 ... 50 s per E-step there).

    python tools/fuzz_fused.py [cases] [seed] [big cases] [max points] [long cases at >= 100k]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpd_c, cpd_numpy as co  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402


def make_clouds(rng, shape, kind, m, n, seed):
    if shape == "surface":
        src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=seed) if kind == "rigid" else synthetic.affine_pair(n, m=m, seed=seed)
        return src, tgt
    g = np.random.default_rng(seed)
    if shape in ("aniso", "volume"):
        box = np.array([10.0, 1.0, 1.0]) if shape == "aniso" else np.array([1.0, 1.0, 1.0])
        base = g.random((max(m, n), 3)) * box
    else:  # clusters
        k = max(m, n)
        a = 0.4 * g.standard_normal((k // 2, 3))
        b = 0.15 * g.standard_normal((k - k // 2, 3)) + np.array([6.0, 0.5, -0.3])
        base = np.concatenate([a, b])[g.permutation(k)]
    src = base[:m].copy()
    lin = synthetic.rot_zx(float(g.uniform(-14, 14)), float(g.uniform(-6, 6)))
    if kind == "affine":
        lin = lin @ np.diag([1.06, 0.96, 1.0]) + 0.03 * np.array([[0, 1, 0], [0, 0, 0], [0, 0, 0]])
    noise = 0.004  # (absolute: the same noise level for every shape)
    # the target is the same sample (another subset when the sizes differ), moved, with noise, in another order
    tgt = (base[:n] @ lin.T + np.array([0.05, -0.03, 0.02]) + noise * g.standard_normal((n, 3)))[g.permutation(n)]
    return src, tgt


def run_case(rng, label, m, n, dim, kind, shape, w, iters, upd, far, with_init, seed):
    src, tgt = make_clouds(rng, shape, kind, m, n, seed)
    if dim == 2:
        src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
    if far:
        off = rng.uniform(100, 1500, dim) * rng.choice([-1.0, 1.0], dim)
        src, tgt = src + off, tgt + off
    init = None
    if kind == "rigid" and with_init:
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
    reg = (cpd.RigidCPD if kind == "rigid" else cpd.AffineCPD)(src, **kw)
    res = reg.registration(tgt, w=w, maxiter=iters, tol=-1.0)
    plan = reg._plan
    how = "single/%s" % ("mfma" if plan.last_estep_engine() else "valu") if plan.last_estep_fused() else \
        "two/%d%d%s" % (plan.last_estep_engines() + ("L" if plan.last_estep_lean() else "",))
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
    amp = float(np.mean(np.sum((tgt - tgt.mean(0)) ** 2, axis=1))) / (dim * s2)
    err = max(e_lin, e_t, 10.0 * e_s)
    flag = "" if err < 1e-4 else "   <-- OUT OF TOLERANCE"
    print("case %-4s m=%6d n=%6d dim=%d %-6s %-8s far=%d w=%.1f scale=%d init=%d it=%2d seed=%6d last %-11s amp %8.0f: lin %.1e t %.1e sigma2 %.1e%s" % (
        label, m, n, dim, kind, shape, far, w, upd, init is not None, iters, seed, how, amp, e_lin, e_t, e_s, flag), flush=True)
    return err, e_s, kind


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    big = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    nmax = int(sys.argv[4]) if len(sys.argv) > 4 else 60000
    worst, worst_s2, bad, t0 = 0.0, {"rigid": 0.0, "affine": 0.0}, 0, time.time()
    shapes = ["surface", "surface", "aniso", "volume", "clusters"]

    def record(r):
        nonlocal worst, bad
        err, e_s, kind = r
        worst = max(worst, err)
        worst_s2[kind] = max(worst_s2[kind], e_s)
        bad += err >= 1e-4

    for c in range(cases):
        m = int(rng.integers(8192, nmax))
        n = int(rng.integers(8192, nmax))
        kind = str(rng.choice(["rigid", "rigid", "affine"]))
        shape = str(rng.choice(shapes))
        dim = 3 if shape != "surface" else int(rng.choice([2, 3, 3]))
        record(run_case(rng, str(c), m, n, dim, kind, shape, float(rng.choice([0.0, 0.0, 0.1, 0.4])), int(rng.integers(1, 33)),
                        bool(rng.integers(0, 2)) if kind == "rigid" else True, rng.random() < 0.3, rng.random() < 0.4,
                        int(rng.integers(0, 10 ** 6))))
    for c in range(big):  # one EM iteration each, continued from ... the identity: the dense regime at sizes beyond C1
        n = int(rng.choice([100000, 160000, 250000]))
        kind = "affine" if c % 2 else "rigid"
        record(run_case(rng, "B%d" % c, n, n, 3, kind, str(rng.choice(["surface", "aniso", "clusters"])), 0.0, 1, True, False, False,
                        int(rng.integers(0, 10 ** 6))))
    # [r6] whole stretches of registrations at >= 100k points: 10 ... 14 EM iterations from the identity (the fused sweep, its masks, the
    # hand-over to the owner sweep at these sizes; affine: both matrix-core sweeps, the lean row pass and the hand-over), shapes and
    # far offsets as above - the oracle's E-steps cost ~6 ... 12 s each on the bench host
    longs = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    for c in range(longs):
        m = int(rng.integers(100000, 130001))
        n = int(rng.integers(100000, 130001))
        kind = "affine" if c % 3 == 2 else "rigid"
        shape = str(rng.choice(shapes))
        record(run_case(rng, "L%d" % c, m, n, 3, kind, shape, float(rng.choice([0.0, 0.0, 0.1])), int(rng.integers(10, 15)),
                        True, rng.random() < 0.3, False, int(rng.integers(0, 10 ** 6))))
    cases += longs
    print("%d cases, %d out of tolerance, worst %.2e (max of transform errors and 10 x sigma2 error; tolerance 1e-4); worst sigma2 error "
          "rigid %.2e affine %.2e (tolerance 1e-5), %.0f s" % (cases + big, bad, worst, worst_s2["rigid"], worst_s2["affine"], time.time() - t0))


if __name__ == "__main__":
    main()
