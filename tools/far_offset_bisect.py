#!/usr/bin/env python
"""Where do the far-offset fuzz cases get their sigma2 error from?  (profiles/r5_fuzz_fused.log: case 48 - rigid, 2-D surface, clouds
~1000 units from the origin, fused matrix-core sweep - 5.3e-6; case 9 - affine volume, far, lean row pass - 4.8e-6; every other
case <= 1.4e-6.)  The product centres both clouds in fp64 and uploads float32; the oracle works on the caller's float64 arrays.

Variants of the same registration, sigma2 and transform against the oracle:
  raw        the fuzz case as it was (far offset, float64 inputs that are NOT float32-representable)
  near       the same clouds without the offset (inputs float32-representable)
  far32      far offset, inputs rounded to float32 first (both paths see the same float32 values; |x| ~ 1000 has 6e-5 granularity)
  cast       far offset; the ORACLE is fed what the GPU sees: (x - centroid) rounded to float32, centroid added back in float64
  valu       raw, matrix cores off (vector-pipe sweeps only)
If `cast` closes the gap, the error is the float32 cast of the centred inputs - the product's input precision, not an engine.

    python tools/far_offset_bisect.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpd_c, cpd_numpy as co  # noqa: E402
from probreg_amd import cpd, synthetic  # noqa: E402


def oracle(kind, src, tgt, iters, w=0.0):
    dim = src.shape[1]
    p = dict(rot=np.identity(dim), t=np.zeros(dim), scale=1.0) if kind == "rigid" else dict(b=np.identity(dim), t=np.zeros(dim))
    s2 = co.squared_kernel_sum_closed_form(src, tgt)
    for _ in range(iters):
        es = co.EstepResult(*cpd_c.expectation_step(co.transform(kind, p, src), tgt, s2, w))
        p, s2, _q = co.mstep_rigid(src, tgt, es) if kind == "rigid" else co.mstep_affine(src, tgt, es)
    return p, s2


def gpu(kind, src, tgt, iters, dense=1):
    reg = (cpd.RigidCPD if kind == "rigid" else cpd.AffineCPD)(src)
    reg._initialize(tgt)
    reg._plan.set_dense_engine(dense)
    res = reg.registration(tgt, maxiter=iters, tol=-1.0)
    return res


def report(tag, kind, res, p, s2):
    lin, want = (res.transformation.rot, p["rot"]) if kind == "rigid" else (res.transformation.b, p["b"])
    print("  %-6s sigma2 rel err %.2e   lin %.1e   t %.1e   (sigma2 %.6e)" % (
        tag, abs(res.sigma2 - s2) / s2, np.max(np.abs(lin - want)), np.max(np.abs(res.transformation.t - p["t"])) / max(1.0, np.max(np.abs(p["t"]))), s2), flush=True)


def cast_like_gpu(a):
    c = a.mean(axis=0)
    return (a - c).astype(np.float32).astype(np.float64) + c


cases = [("rigid", 2, 31730, 41596, 311601, 10, "surface"), ("affine", 3, 20000, 26000, 777, 12, "volume"),
         ("rigid", 3, 30000, 30000, 5, 16, "surface")]
rng = np.random.default_rng(12)
for kind, dim, m, n, seed, iters, shape in cases:
    if shape == "surface":
        src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=seed) if kind == "rigid" else synthetic.affine_pair(n, m=m, seed=seed)
    else:
        g = np.random.default_rng(seed)
        base = g.random((max(m, n), 3))
        lin = synthetic.rot_zx(9.0, -4.0) @ np.diag([1.06, 0.96, 1.0])
        src = base[:m].copy()
        tgt = (base[:n] @ lin.T + np.array([0.05, -0.03, 0.02]) + 0.004 * g.standard_normal((n, 3)))[g.permutation(n)]
    src, tgt = src[:, :dim].copy(), tgt[:, :dim].copy()
    off = rng.uniform(100, 1500, dim) * rng.choice([-1.0, 1.0], dim)
    print("%s %d-D %s m=%d n=%d, %d iterations, offset %s" % (kind, dim, shape, m, n, iters, np.round(off, 1)))
    fs, ft = src + off, tgt + off
    p, s2 = oracle(kind, fs, ft, iters)
    report("raw", kind, gpu(kind, fs, ft, iters), p, s2)
    report("valu", kind, gpu(kind, fs, ft, iters, dense=0), p, s2)
    pn, s2n = oracle(kind, src, tgt, iters)
    report("near", kind, gpu(kind, src, tgt, iters), pn, s2n)
    f32s, f32t = fs.astype(np.float32).astype(np.float64), ft.astype(np.float32).astype(np.float64)
    p3, s23 = oracle(kind, f32s, f32t, iters)
    report("far32", kind, gpu(kind, f32s, f32t, iters), p3, s23)
    pc, s2c = oracle(kind, cast_like_gpu(fs), cast_like_gpu(ft), iters)
    report("cast", kind, gpu(kind, fs, ft, iters), pc, s2c)
