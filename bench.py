#!/usr/bin/env python
"""Benchmark of the CPD / FilterReg EM hot path on MI355X.

    python bench.py --gpus N --steps 20 --warmup 3        (N > 1 without WORLD_SIZE in the environment: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one EM iteration (transform + E-step + fp64 moment reduction + [all-reduce] + device M-step) of RigidCPD on
BASELINE.json's config C1 - the E-step as the registration's own loop runs it: ONE sweep over the pairs per iteration in every regime
(the rigid M-step's moments from per-column sums: the fused matrix-core sweep while sigma2 is large, DESIGN.md 3.1e, the residual-form
sweep of the column blocks' owners on the vector pipe afterwards, 3.1f / 3.1g; PROBREG_BENCH_TWO_SWEEPS=1 keeps the column pass +
row pass): synthetic N = M = 100 000 3-D points, fp32 pair arithmetic, w = 0.  With N GPUs the SAME problem is solved with the target cloud sharded over the ranks ("scaling":
"strong"); one all-reduce of 32 doubles per iteration.  Inputs are resident in HBM before the timed region.

THE TIMED WINDOW IS PINNED: the E-step's cost depends on sigma2 (the sweeps skip blocks whose every pair is an exact
fp32 zero), so after the W warm-up steps the EM state is put back to the start of the registration and the K timed
steps are EM iterations 0 .. K-1 of that registration - `--warmup` cannot move the headline.  The line also carries
`dense_it_s` (iteration 0 alone: every pair evaluated), `late_it_s` (iterations 45..49) and `trajectory_it_s` (= value).

Rank 0 prints ONE JSON line.  With the defaults (1 GPU, C1) it also carries
  roofline        binding roofline of the dominant kernel (row pass): fp32 VALU, from the pairs the kernel actually
                  evaluated (device counters) x its flop per pair, HIP-event timed on the plan's stream
  parity          GPU vs the CPU oracle on the cpu_baseline sample (same inputs, same iterations)
  cpu_baseline    oracle/cpd_estep_c.c timed on the host cores at three sizes, fitted t = a M N
  other_workloads C2 (affine 200k), C3 (non-rigid 50k), C4 (FilterReg 500k) measured the same way, each with its roofline
                  and its own GPU-vs-oracle parity block (reduced samples of the same generators)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E 8 TB/s
VALU_F32_PEAK_TFLOPS = 157.3   # 256 CU x 4 SIMD x 32 lanes x 2 flop (FMA) x 2.4 GHz
F64_MFMA_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64

# Flop per EVALUATED source-target pair, counted from the vector-pipe sweeps' instruction mix (fma = 2, else = 1):
#   row pass   3 sub + 3 fma (d2) + 1 fma (exponent) + 1 exp + 1 add (p1) + 4 fma (u, e)            = 21
#   col pass   3 sub + 3 fma (d2) + 1 min + 1 fma (exponent) + 1 exp + 1 add (sum)                  = 14
# The same count is charged to the dense-regime iterations that take the exponent (8 of the 21 / 14 flop) from the bf16
# matrix pipe instead (csrc/cpd_sweeps_mfma.hip): `frac` is useful pair-flops per second over the fp32 vector peak.
FLOP_ROW, FLOP_COL = 21.0, 14.0
# the fused single sweep of a rigid iteration in the dense regime (prg_cpd_set_moments_only): the column pass'
# exponent, minimum and sum (14) + the contraction of 4 more channels on the source side (4 fma = 8): ONE sweep per E-step
FLOP_FUSED = 22.0
# the matrix-core row pass without its residual sums (prg_cpd_last_estep_lean): one fma per pair less
FLOP_ROW_LEAN = 19.0
# SURVEY.md 8(d)'s own count for the two-sweep fused form (scale as a separate multiplication, no min tracking): 20 / 11
FLOP_ROW_SURVEY, FLOP_COL_SURVEY = 20.0, 11.0

WORKLOADS = {
    # name: (kind, N = M, description)
    "rigid_100k": ("rigid", 100000, "C1 RigidCPD fp32 synthetic N=M=100000 D=3 w=0"),
    "affine_200k": ("affine", 200000, "C2 AffineCPD fp32 synthetic N=M=200000 D=3 w=0"),
    "nonrigid_50k": ("nonrigid", 50000, "C3 NonRigidCPD fp32 E-step / fp64 M-step synthetic N=M=50000 beta=2 lmd=2"),
    "filterreg_500k": ("filterreg", 500000, "C4 FilterReg rigid pt2pt synthetic N=M=500000, 5% outliers, w=0.05"),
    "bcpd_20k": ("bcpd", 20000, "BCPD (SURVEY 8f rank 4) fp32 E-step / fp64 M-step synthetic N=M=20000 lmd=2 w=0.05"),
    "rigid_20k": ("rigid", 20000, "reduced RigidCPD fp32 synthetic N=M=20000 (debug only)"),
}

# the UNMODIFIED reference NumPy path (stub import) probed at survey time on the 8-vCPU build
# container - it cannot run on the GPU box (/root/reference does not exist there), so it rides along as a constant.
REFERENCE_NUMPY_PROBE = {
    "what": "reference probreg RigidCPD NumPy path, EM iterations/s (maxiter=20, tol=0), 8 vCPU Xeon 2.1 GHz (BASELINE.md 2)",
    "it_s_at_n_eq_m": {"1000": 19.95, "3000": 5.10, "6000": 0.99},
    "extrapolated_s_per_iteration_at_100k": 225.0,
}


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle, rank 0 / N = 1 only) and the parity block that rides on the same sample
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline_and_parity(n_full, kind="rigid"):
    """C oracle timed at three sizes (fit t_iter = a M N) + GPU-vs-oracle parity on the largest sample."""
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import cpd, synthetic

    # [r6] 30k / 45k / 60k after one untimed call (the OpenMP pool's start-up sat in the 10k point of round 5's fit: +71 %; a 20k
    # point - 0.4 s of work on 128 threads - still read +12 ... +32 % from box to box)
    sizes, iters = (30000, 45000, 60000), 2
    per_iter, last = [], None
    w_src, w_tgt, _ = synthetic.rigid_pair(4000, seed=0)
    cpd_c.expectation_step(w_src, w_tgt, co.squared_kernel_sum_closed_form(w_src, w_tgt), 0.0)
    for ns in sizes:
        src, tgt, _ = synthetic.rigid_pair(ns, seed=0)
        sigma2 = co.squared_kernel_sum_closed_form(src, tgt)
        params = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0)
        t0 = time.perf_counter()
        for _ in range(iters):
            ts = co.transform("rigid", params, src)
            es = co.EstepResult(*cpd_c.expectation_step(ts, tgt, sigma2, 0.0))
            params, sigma2, _q = co.mstep_rigid(src, tgt, es)
        per_iter.append((time.perf_counter() - t0) / iters)
        last = (src, tgt, params, sigma2)
    mn = np.array([float(s) * s for s in sizes])
    t = np.array(per_iter)
    a = float(np.sum(t * mn) / np.sum(mn * mn))  # least squares through the origin
    resid = [float(ti / (a * x) - 1.0) for ti, x in zip(t, mn)]
    baseline = {
        "value": 1.0 / (a * float(n_full) * n_full),
        "unit": "EM iterations/s",
        "cores": cpd_c.threads(),
        "kind": "port",
        "sample": "oracle/cpd_estep_c.c (C/OpenMP fp64 restatement of probreg cpd.py:71-88) + numpy M-step, RigidCPD "
                  "N=M in %s, %d iterations each; fit t_iter = a*M*N, value = 1/(a*%d^2) (extrapolated)"
                  % (list(sizes), iters, n_full),
        "fit": {"a_seconds_per_pair": a, "sizes": list(sizes), "s_per_iteration": per_iter,
                "relative_residuals": resid},
        "reference_numpy_probe": REFERENCE_NUMPY_PROBE,
    }
    baseline["reference_numpy"] = reference_numpy_baseline(n_full)
    # parity: the product on the SAME input (the largest sample) for the SAME number of iterations
    src, tgt, p, s2 = last
    res = cpd.registration_cpd(src, tgt, "rigid", maxiter=iters, tol=-1.0)
    tr = res.transformation
    parity = {
        "against": "oracle (C E-step + numpy M-step, fp64) on synthetic.rigid_pair(%d, seed=0), %d EM iterations"
                   % (sizes[-1], iters),
        "rot_max_abs_err": float(np.max(np.abs(tr.rot - p["rot"]))),
        "t_max_abs_err": float(np.max(np.abs(tr.t - p["t"]))),
        "scale_rel_err": float(abs(tr.scale - p["scale"]) / abs(p["scale"])),
        "sigma2_rel_err": float(abs(res.sigma2 - s2) / s2),
        "tolerance": {"transform": 1e-4, "sigma2": 1e-5},
        "full_size": "tests/test_fullsize_gpu.py holds C1 (100k, dense + late regime), C2 (200k), C3 (12k) and C4 (500k) "
                     "to the oracle at the same tolerances",
    }
    parity["ok"] = bool(parity["rot_max_abs_err"] < 1e-4 and parity["t_max_abs_err"] < 1e-4
                        and parity["scale_rel_err"] < 1e-4 and parity["sigma2_rel_err"] < 1e-5)
    return baseline, parity


def reference_numpy_baseline(n_full, sizes=(10000, 20000, 30000), iters=2):
    """probreg's own NumPy formulation of the EM iteration (cpd.py:71-88 as ONE dense M x N float64 matrix: scipy cdist,
    exp, divide, sums, dot - oracle/cpd_numpy.expectation_step_unchunked, held to the reference's outputs at 1e-12 by
    tests/test_oracle_golden.py - plus the M-step of cpd.py:160-192) timed on THIS host at three sizes the matrix fits,
    fitted t_iter = a M N and extrapolated to the bench size.  NumPy / BLAS use the threads they find (all cores)."""
    from oracle import cpd_numpy as co
    from probreg_amd import synthetic

    per_iter = []
    # [r5] sizes at which the M x N passes dominate (at 2k / 5k the per-call overheads put the fit's residuals at +140 % / +34 %):
    # 10k / 20k / 30k - 7.2 GB per float64 matrix at 30k, a handful of them alive at once; a host without the room keeps to what fits
    try:
        import psutil

        free_gb = psutil.virtual_memory().available / 2.0 ** 30
        sizes = tuple(ns for ns in sizes if 5.0 * 8.0 * ns * ns / 2.0 ** 30 < 0.6 * free_gb) or (2000, 5000, 10000)
    except Exception:
        pass
    src, tgt, _ = synthetic.rigid_pair(500, seed=0)
    co.expectation_step_unchunked(src, tgt, 1.0, 0.0)  # imports, thread pools
    for ns in sizes:
        src, tgt, _ = synthetic.rigid_pair(ns, seed=0)
        sigma2 = co.squared_kernel_sum_closed_form(src, tgt)
        params = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0)
        t0 = time.perf_counter()
        for _ in range(iters):
            es = co.expectation_step_unchunked(co.transform("rigid", params, src), tgt, sigma2, 0.0)
            params, sigma2, _q = co.mstep_rigid(src, tgt, es)
        per_iter.append((time.perf_counter() - t0) / iters)
    mn = np.array([float(s) * s for s in sizes])
    t = np.array(per_iter)
    a = float(np.sum(t * mn) / np.sum(mn * mn))
    return {
        "value": 1.0 / (a * float(n_full) * n_full), "unit": "EM iterations/s", "kind": "reference formulation (numpy / scipy, fp64)",
        "cores": os.cpu_count(),
        "sample": "RigidCPD N=M in %s, %d EM iterations each, dense M x N float64 matrix as in probreg cpd.py:71-88; fit t_iter = a*M*N, "
                  "value = 1/(a*%d^2) (extrapolated: the 100k matrix alone would be 80 GB and the reference holds three of them)"
                  % (list(sizes), iters, n_full),
        "fit": {"a_seconds_per_pair": a, "sizes": list(sizes), "s_per_iteration": per_iter,
                "it_s_measured": [1.0 / x for x in per_iter],
                "relative_residuals": [float(ti / (a * x) - 1.0) for ti, x in zip(t, mn)]},
    }


def parity_other_workloads():
    """GPU vs CPU oracle for C2 / C3 / C4 on reduced samples of the same generators (the C1 block rides on the
    cpu_baseline sample); the full-size comparisons live in tests/test_fullsize_gpu.py.  Tolerances: BASELINE.json."""
    from oracle import cpd_c, cpd_numpy as co, filterreg_numpy as fo
    from probreg_amd import cpd, filterreg, synthetic

    out = {}
    tol = {"transform": 1e-4, "sigma2": 1e-5}
    # C2: affine, 40k x 40k, 2 iterations (C E-step + numpy M-step)
    n, iters = 40000, 2
    src, tgt, _ = synthetic.affine_pair(n, seed=0)
    s2 = co.squared_kernel_sum_closed_form(src, tgt)
    p = dict(b=np.identity(3), t=np.zeros(3))
    for _ in range(iters):
        es = co.EstepResult(*cpd_c.expectation_step(co.transform("affine", p, src), tgt, s2, 0.0))
        p, s2, _q = co.mstep_affine(src, tgt, es)
    res = cpd.registration_cpd(src, tgt, "affine", maxiter=iters, tol=-1.0)
    e = {"against": "oracle (C E-step + numpy M-step, fp64) on synthetic.affine_pair(%d, seed=0), %d EM iterations" % (n, iters),
         "b_rel_err": float(np.max(np.abs(res.transformation.b - p["b"])) / np.max(np.abs(p["b"]))),
         "t_max_abs_err": float(np.max(np.abs(res.transformation.t - p["t"]))),
         "sigma2_rel_err": float(abs(res.sigma2 - s2) / s2), "tolerance": tol}
    e["ok"] = bool(e["b_rel_err"] < 1e-4 and e["t_max_abs_err"] < 1e-4 and e["sigma2_rel_err"] < 1e-5)
    out["affine_200k"] = e
    # C3: non-rigid, 4k x 4k (numpy LAPACK solve on the float32 G), 3 iterations
    n, iters = 4000, 3
    src, tgt = synthetic.nonrigid_pair(n, seed=0)
    p, s2, _q, _ = co.registration("nonrigid", src, tgt, maxiter=iters, tol=-1.0, closed_form_init=True)
    want = co.transform("nonrigid", p, src, co.rbf_kernel(src, src, 2.0))
    res = cpd.registration_cpd(src, tgt, "nonrigid", maxiter=iters, tol=-1.0)
    got = res.transformation.transform(src)
    e = {"against": "oracle (numpy fp64, reference formulation cpd.py:284-303 on the float32 G) on "
                    "synthetic.nonrigid_pair(%d, seed=0), %d EM iterations" % (n, iters),
         "transformed_source_max_err_over_extent": float(np.max(np.abs(got - want)) / np.max(np.abs(want - want.mean(0)))),
         "sigma2_rel_err": float(abs(res.sigma2 - s2) / s2), "tolerance": tol,
         "full_size": "tests/test_fullsize_gpu.py::test_nonrigid_c3_full_size_vs_lapack (N = M = 50 000, LAPACK dgesv on the host)"}
    e["ok"] = bool(e["transformed_source_max_err_over_extent"] < 1e-4 and e["sigma2_rel_err"] < 1e-5)
    out["nonrigid_50k"] = e
    # C4: FilterReg, 50k x 50k with 5 % outliers, 4 iterations (C lattice, bit-identical to the reference's vendored one)
    n, iters = 50000, 4
    src, tgt, _ = synthetic.filterreg_pair(n, seed=0)
    s2_0 = float(np.float32(co.squared_kernel_sum_closed_form(src, tgt)))
    rot, t, s2, q, _k = fo.registration(src, tgt, sigma2=s2_0, update_sigma2=True, w=0.05, maxiter=iters, tol=-1.0)
    res = filterreg.registration_filterreg(src, tgt, sigma2=s2_0, update_sigma2=True, w=0.05, maxiter=iters, tol=-1.0)
    e = {"against": "oracle (filterreg_numpy on the C permutohedral lattice) on synthetic.filterreg_pair(%d, seed=0), "
                    "%d EM iterations, w = 0.05, sigma2 updated" % (n, iters),
         "rot_max_abs_err": float(np.max(np.abs(res.transformation.rot - rot))),
         "t_max_abs_err": float(np.max(np.abs(res.transformation.t - t))),
         "sigma2_rel_err": float(abs(res.sigma2 - s2) / s2), "q_rel_err": float(abs(res.q - q) / abs(q)), "tolerance": tol}
    e["ok"] = bool(e["rot_max_abs_err"] < 1e-4 and e["t_max_abs_err"] < 1e-4 and e["sigma2_rel_err"] < 1e-5)
    out["filterreg_500k"] = e
    return out


def _pmc_traffic(workload, key):
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(path):
        try:
            return json.load(open(path)).get(workload, {}).get(key)
        except Exception:
            return None
    return None


def _base(metric, value, elapsed, steps, warmup, desc, dtype, world=1):
    return {"metric": metric, "value": value, "unit": "EM iterations/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc}}


# ----------------------------------------------------------------------------------------------------------------
# C1 / C2: rigid / affine CPD
# ----------------------------------------------------------------------------------------------------------------
def bench_cpd(workload, steps, warmup, tuning="", pairs_log=None):
    import torch
    from probreg_amd import _lib, cpd, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    kind, n, desc = WORKLOADS[workload]
    if kind == "rigid":
        src, tgt, truth = synthetic.rigid_pair(n, seed=0)
        reg = cpd.RigidCPD(src)
        kind_id = _lib.PRG_TF_RIGID
    else:
        src, tgt, truth = synthetic.affine_pair(n, seed=0)
        reg = cpd.AffineCPD(src)
        kind_id = _lib.PRG_TF_AFFINE
    reg._initialize(tgt)  # upload (target rows sharded over ranks), sigma2 initialiser
    plan = reg._plan
    if tuning:
        plan.set_tuning(*[int(v) for v in tuning.split(",")])
    if kind == "rigid" and os.environ.get("PROBREG_BENCH_TWO_SWEEPS") != "1":
        # what the registration's own loop does (prg_cpd_iterate): a rigid E-step feeds nothing but the rigid M-step, so the
        # dense regime may run as ONE sweep over the pairs; the step below stays E-step [+ all-reduce] + M-step
        plan.set_moments_only(1)

    def step():
        plan.estep(0.0)
        reg._all_reduce_moments(plan)
        plan.mstep(kind_id, True)

    def fence():
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(warmup):
        step()
    reg._restart()                 # EM state back to iteration 0: the timed window is iterations 0 .. steps-1
    elapsed = timed(steps)
    res = reg._result_from_params(plan.get_params())

    # regimes: iteration 0 alone (every pair evaluated) and iterations 45..49 (sigma2 ~ noise level: ~97 % culled)
    reg._restart()
    t_dense = timed(1)
    for _ in range(44):
        step()
    t_late = timed(5) / 5.0

    # Per-kernel timing with HIP events on the plan's stream + the device counters of evaluated pairs: the SAME
    # iterations 0 .. steps-1 replayed (their duration depends on sigma2, so they are averaged over the trajectory
    # the timed region walked).
    reg._restart()
    acc, first, last = {}, None, None
    pairs_row = pairs_col = flop_row = 0.0
    first_pairs = last_pairs = None
    per_iteration = []
    for it in range(steps):
        s2_it = float(plan.get_params()[13])
        ms = plan.estep_timed(0.0)
        pc, pr = plan.pair_counts()
        ce, re_ = plan.last_estep_engines()
        fused = bool(plan.last_estep_fused())
        fr = FLOP_ROW_LEAN if plan.last_estep_lean() else FLOP_ROW
        if fused:  # one sweep did the whole E-step: it is reported in the row-pass slot (the dominant kernel), the column slot is empty
            ms = dict(ms, rowpass=ms["colpass"], colpass=0.0)
            pr, pc, fr, re_ = pc, 0.0, FLOP_FUSED, 2
        flop_row += pr * fr
        per_iteration.append((it, s2_it, ce, re_, pc, pr, ms["colpass"], ms["rowpass"], ms["total"], fr))
        reg._all_reduce_moments(plan)
        plan.mstep(kind_id, True)
        if first is None:
            first, first_pairs, first_fr = dict(ms), (pc, pr), fr
        last, last_pairs, last_fr = dict(ms), (pc, pr), fr
        pairs_col += pc
        pairs_row += pr
        for k, v in ms.items():
            acc[k] = acc.get(k, 0.0) + v / steps
    if rank != 0:
        return None
    if pairs_log:  # what `roofline.frac` is made of, iteration by iteration (tools/profile_round.sh commits it)
        with open(pairs_log, "w") as f:
            f.write("# %s: E-step sweeps of the timed window, HIP events on the plan's stream (prg_cpd_estep_timed) and the device's "
                    "counters of evaluated pairs (prg_cpd_pair_counts)\n" % desc)
            f.write("# row = 2: a SINGLE-sweep E-step of a rigid iteration (col 1: the fused matrix-core sweep, col 0: the residual-form vector-pipe sweep; %g flop per pair, reported in the row columns; the column slot is empty)\n" % FLOP_FUSED)
            f.write("# engine 1 = matrix cores (bf16x3 exponent, csrc/cpd_sweeps_mfma.hip), 0 = culled vector-pipe sweeps; "
                    "frac = pairs x flop/pair / ms / %.1f TFLOP/s (flop/pair: row %g - %g where the matrix-core row pass ran "
                    "without its residual sums, column `fr` - column %g)\n"
                    % (VALU_F32_PEAK_TFLOPS, FLOP_ROW, FLOP_ROW_LEAN, FLOP_COL))
            f.write("%3s %12s %4s %4s %3s %14s %14s %9s %9s %9s %8s %8s\n" % ("it", "sigma2", "col", "row", "fr", "pairs_col", "pairs_row",
                                                                           "ms_col", "ms_row", "ms_estep", "frac_col", "frac_row"))
            for (it, s2_it, ce, re_, pc, pr, mc, mr, mt, fr) in per_iteration:
                f.write("%3d %12.5e %4d %4d %3.0f %14.0f %14.0f %9.4f %9.4f %9.4f %8.3f %8.3f\n" % (
                    it, s2_it, ce, re_, fr, pc, pr, mc, mr, mt, (pc * FLOP_COL / (mc * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS) if mc > 0 else 0.0,
                    pr * fr / (mr * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS))
            tot_r = sum(r[5] for r in per_iteration)
            tot_fl = sum(r[5] * r[9] for r in per_iteration)
            tot_mr = sum(r[7] for r in per_iteration)
            f.write("# window: row pass %.6e pairs, %.6e flop in %.4f ms -> %.2f TFLOP/s = frac %.4f\n" % (
                tot_r, tot_fl, tot_mr, tot_fl / (tot_mr * 1e-3) / 1e12, tot_fl / (tot_mr * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS))

    m_pts, n_loc = plan.m, plan.n
    row_s_total, col_s_total = acc["rowpass"] * 1e-3 * steps, acc["colpass"] * 1e-3 * steps
    row_tf = flop_row / row_s_total / 1e12
    col_tf = pairs_col * FLOP_COL / col_s_total / 1e12 if col_s_total > 0 else 0.0
    dense_row_tf = first_pairs[1] * first_fr / (first["rowpass"] * 1e-3) / 1e12
    dense_col_tf = first_pairs[0] * FLOP_COL / (first["colpass"] * 1e-3) / 1e12 if first["colpass"] > 0 else 0.0
    # SURVEY.md 8(d) figure kept beside it: algorithmic bytes of the reference's formulation at fp32 (P written once
    # by the column pass, read once by the row pass) over the measured time - an EFFECTIVE rate, not a roofline
    alg_row = 4.0 * m_pts * n_loc + 4.0 * (m_pts + n_loc) * 5
    out = _base("EM iterations/sec (RigidCPD, N=M=100k fp32)" if workload == "rigid_100k"
                else "EM iterations/sec (%s)" % desc, steps / elapsed, elapsed, steps, warmup, desc, "f32", world)
    out["config"].update({
        "window": "EM iterations 0..%d of the registration (state reset after the warm-up steps)" % (steps - 1),
        "target_sharding": "cells of a recursive bisection of the target (probreg_amd.dist.bisection_shards) over %d rank(s)" % world,
        "collective": (("1 ncclAllReduce (RCCL, issued by libprobreg_hip.so on the plan's stream) of the moment block's 24 fp64 sums per iteration"
                        if getattr(plan, "_comm", None) is not None else
                        "1 torch.distributed all_reduce (%s) of 32 fp64 per iteration" % torch.distributed.get_backend())
                       if world > 1 else "none"),
        "collective_path": ("library-side RCCL (prg_cpd_set_comm: ncclAllReduce inside prg_cpd_estep, plan's stream)"
                            if getattr(plan, "_comm", None) is not None else
                            "torch.distributed (%s)" % torch.distributed.get_backend()
                            if torch.distributed.is_available() and torch.distributed.is_initialized() else "none: one process"),
        "m": m_pts, "n_local": n_loc, "n_global": n})
    out["trajectory_it_s"] = steps / elapsed
    out["dense_it_s"] = 1.0 / t_dense
    out["late_it_s"] = 1.0 / t_late
    out["roofline"] = {
        "bound": "valu",
        "kernel": "the E-step's dominant pair sweep: rigid - ONE sweep per iteration in every regime (the fused matrix-core sweep "
                  "k_colpass_mfma<FUSED> while sigma2 is large, the residual-form vector-pipe sweep run by the column blocks' owners, "
                  "k_colpass_owner, afterwards; den, P1, PX sums from the column side, one exponential per pair and iteration); "
                  "affine - the row pass (k_rowpass_mfma / k_rowpass_queue) of its two sweeps",
        "achieved": row_tf,
        "peak": VALU_F32_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": row_tf / VALU_F32_PEAK_TFLOPS,
        "traffic": _pmc_traffic(workload, "dominant_sweep_hbm_bytes_per_launch") or _pmc_traffic(workload, "rowpass_hbm_bytes_per_launch"),
        "traffic_source": "profiles/pmc_traffic.json - STATIC: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                          "tools/profile_round.sh, (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; not measured in this run",
        "how": "flop = pairs the kernel evaluated (per-workgroup device counters, prg_cpd_pair_counts) x %g flop/pair (%g in the "
               "iterations whose matrix-core row pass ran without its residual sums, prg_cpd_last_estep_lean; %g in the iterations "
               "that ran as the fused single sweep, prg_cpd_last_estep_fused - there the sweep is the whole E-step); "
               "time = HIP events on the plan's stream; both summed over the %d timed-window iterations" % (FLOP_ROW, FLOP_ROW_LEAN, FLOP_FUSED, steps),
        "flop_per_pair": {"isa_count": {"row": FLOP_ROW, "row_lean": FLOP_ROW_LEAN, "col": FLOP_COL, "fused": FLOP_FUSED,
                                        "row_window_average": flop_row / max(pairs_row, 1.0)},
                          "survey_8d": {"row": FLOP_ROW_SURVEY, "col": FLOP_COL_SURVEY},
                          "note": "the matrix-core engine takes 8 of these flop per pair (the exponent) from the bf16 matrix "
                                  "pipe; frac charges them all to the fp32 vector peak = useful pair-flops per second"},
        "frac_with_survey_flops": pairs_row * FLOP_ROW_SURVEY / row_s_total / 1e12 / VALU_F32_PEAK_TFLOPS,
        "matrix_core_share": {"column_pass_iterations": sum(1 for r in per_iteration if r[2]) / float(steps),
                              "row_pass_iterations": sum(1 for r in per_iteration if r[3]) / float(steps),
                              "row_pass_pairs": sum(r[5] for r in per_iteration if r[3]) / max(pairs_row, 1.0),
                              "row_pass_lean_iterations": sum(1 for r in per_iteration if r[9] == FLOP_ROW_LEAN) / float(steps),
                              "fused_single_sweep_iterations": sum(1 for r in per_iteration if r[3] == 2 and r[2]) / float(steps),
                              "residual_form_single_sweep_iterations": sum(1 for r in per_iteration if r[3] == 2 and not r[2]) / float(steps)},
        "avg_launch_ms": acc["rowpass"],
        "pairs_evaluated_per_launch": pairs_row / steps,
        "pairs_total_per_launch": float(m_pts) * n_loc,
        "evaluated_fraction": pairs_row / steps / (float(m_pts) * n_loc),
        "dense_regime": {"what": "iteration 0: sigma2 large, every (wave, group) block evaluated",
                         "rowpass": {"ms": first["rowpass"], "pairs": first_pairs[1], "achieved": dense_row_tf,
                                     "frac": dense_row_tf / VALU_F32_PEAK_TFLOPS},
                         "colpass": {"ms": first["colpass"], "pairs": first_pairs[0], "achieved": dense_col_tf,
                                     "frac": dense_col_tf / VALU_F32_PEAK_TFLOPS}},
        "late_regime": {"what": "iteration %d of the window" % (steps - 1),
                        "rowpass": {"ms": last["rowpass"], "pairs": last_pairs[1],
                                    "frac": last_pairs[1] * last_fr / (last["rowpass"] * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS},
                        "colpass": {"ms": last["colpass"], "pairs": last_pairs[0],
                                    "frac": (last_pairs[0] * FLOP_COL / (last["colpass"] * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS)
                                    if last["colpass"] > 0 else 0.0}},
        "colpass": {"achieved": col_tf, "frac": col_tf / VALU_F32_PEAK_TFLOPS, "avg_launch_ms": acc["colpass"],
                    "pairs_evaluated_per_launch": pairs_col / steps, "flop_per_pair": FLOP_COL},
        "effective_hbm": {"what": "SURVEY 8(d) algorithmic bytes (P written once + read once at fp32 = 4 M N per sweep "
                                  "launch) / measured time; the fused sweeps never store P, so this is an effective "
                                  "rate to set beside the 8 TB/s HBM peak, not a fraction of a binding roofline",
                          "algorithmic_bytes_per_launch": alg_row,
                          "rowpass_GBs": alg_row / (acc["rowpass"] * 1e-3) / 1e9,
                          "e_step_GBs": 2.0 * alg_row / (acc["total"] * 1e-3) / 1e9,
                          "hbm_peak_GBs": HBM_PEAK_GBS},
    }
    out["kernel_ms"] = {"mean_over_timed_iterations": acc, "first_timed_iteration": first, "last_timed_iteration": last}
    out["result"] = {"sigma2": res.sigma2, "q": res.q}
    if kind == "rigid":
        r_true, _t_true, _ = truth
        out["result"]["rot_err_vs_truth"] = float(np.max(np.abs(res.transformation.rot - r_true)))
    return out


# ----------------------------------------------------------------------------------------------------------------
# C3: non-rigid CPD
# ----------------------------------------------------------------------------------------------------------------
def bench_nonrigid(workload, steps, warmup, dense_compare=True):
    """One step = E-step (fp32 sweeps over all M x N pairs) + M-step (cpd.py:284-303) on the low-rank factor of G
    (G = F F^T, pivoted Cholesky at set_source).  The timed window is EM iterations 0..steps-1 of the
    registration (state reset after the warm-up).  `dense_solver` times the same M-step through the M x M fp64
    Cholesky the plan falls back to when G is not low rank, and how far the two solutions are apart."""
    import numpy as np
    import torch
    from probreg_amd import cpd, synthetic

    _kind, n, desc = WORKLOADS[workload]
    src, tgt = synthetic.nonrigid_pair(n, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reg = cpd.NonRigidCPD(src)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    reg._initialize(tgt)
    plan = reg._plan
    rank = plan.nonrigid_rank()

    def step():
        plan.estep(0.0)
        plan.mstep_nonrigid(2.0)

    def restart():
        reg._restart()
        plan.set_w(np.zeros_like(src))

    for _ in range(warmup):
        step()
    restart()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sigma2_end = float(plan.get_params()[13])
    # split of one iteration: E-step kernels by HIP events, M-step by synchronised wall time
    restart()
    ms = plan.estep_timed(0.0)
    col_pairs, row_pairs = plan.pair_counts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.mstep_nonrigid(2.0)
    torch.cuda.synchronize()
    t_m = time.perf_counter() - t0
    w_lr = plan.get_w()
    t_lr = plan.nonrigid_apply()
    out = _base("EM iterations/sec (%s)" % desc, steps / elapsed, elapsed, steps, warmup, desc, "f32 E-step / f64 M-step")
    out["config"]["window"] = "EM iterations 0..%d of the registration (state reset after the warm-up steps)" % (steps - 1)
    row_tf = row_pairs * (FLOP_ROW_LEAN if plan.last_estep_lean() else FLOP_ROW) / (ms["rowpass"] * 1e-3) / 1e12
    out["roofline"] = {"bound": "valu", "kernel": "k_rowpass (E-step sweep 2 over all M x N pairs; the M-step is no longer "
                       "the dominant kernel: G = F F^T, rank %d)" % rank,
                       "achieved": row_tf, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": row_tf / VALU_F32_PEAK_TFLOPS,
                       "traffic": _pmc_traffic(workload, "rowpass_hbm_bytes_per_launch"),
                       "avg_launch_ms": ms["rowpass"], "pairs_evaluated_per_launch": row_pairs}
    out["kernel_ms"] = {"estep": {k: float(v) for k, v in ms.items()}, "mstep_wall": 1e3 * t_m}
    out["kernel_factor"] = {"rank": rank, "build_ms_incl_upload": 1e3 * t_build,
                            "what": "pivoted Cholesky of G, columns evaluated on the fly in fp64, stops at 1e-14 per entry"}
    out["result"] = {"sigma2": sigma2_end}
    if rank > 0 and dense_compare:  # the dense fallback on the same E-step, for the record
        try:
            class _Dense(cpd.NonRigidCPD):
                _solver_mode = 0

            del reg, plan
            torch.cuda.synchronize()
            reg = _Dense(src)
            reg._initialize(tgt)
            plan = reg._plan
            plan.estep(0.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.mstep_nonrigid(2.0)
            torch.cuda.synchronize()
            t_d = time.perf_counter() - t0
            flops = n ** 3 / 3.0 + 3 * 2.0 * n * n * 3   # Cholesky + three G-times-(M x 3) products
            t_dense = plan.nonrigid_apply()
            ext = float(np.max(np.abs(t_dense - t_dense.mean(0))))
            out["dense_solver"] = {"what": "same first M-step through the M x M fp64 Cholesky on the float32 G (k_gemm_nt_f64)",
                                   "mstep_ms": 1e3 * t_d, "achieved_TFLOPs": flops / t_d / 1e12,
                                   "frac_of_f64_mfma_peak": flops / t_d / 1e12 / F64_MFMA_PEAK_TFLOPS,
                                   "speedup_of_the_factor": t_d / t_m,
                                   "max_dT_over_extent": float(np.max(np.abs(t_lr - t_dense))) / ext,
                                   "max_dW_rel": float(np.max(np.abs(w_lr - plan.get_w())) / np.max(np.abs(w_lr)))}
        except Exception as e:
            out["dense_solver"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def bench_bcpd(workload, steps, warmup):
    """BCPD: one step = weighted E-step + Woodbury M-step (bcpd.py:82-98 loop body without the convergence test)."""
    import torch
    from probreg_amd import bcpd, synthetic

    _kind, n, desc = WORKLOADS[workload]
    src, tgt = synthetic.nonrigid_pair(n, seed=0)
    src, tgt = src * 10.0, tgt * 10.0  # object ~20 units across: the c = 1 inverse-multiquadric kernel has a sensible width
    reg = bcpd.CombinedBCPD(src)
    stamps = []
    reg.set_callbacks([lambda tr: (torch.cuda.synchronize(), stamps.append(time.perf_counter()))])
    reg.registration(tgt, w=0.05, maxiter=warmup + steps, tol=-1.0)
    elapsed = stamps[-1] - stamps[warmup - 1] if warmup > 0 else stamps[-1] - stamps[0]
    nsteps = steps if warmup > 0 else steps - 1
    plan = reg._plan
    nu, resid = np.ones(n), np.zeros((n, 3))
    plan.bcpd_solve(2.0, 10.0, resid, nu)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.bcpd_solve(2.0, 10.0, resid, nu)
    torch.cuda.synchronize()
    t_m = time.perf_counter() - t0
    flops = 4.0 / 3.0 * n ** 3  # Cholesky (1/3) + triangular solve with M right-hand sides (1)
    out = _base("EM iterations/sec (%s)" % desc, nsteps / elapsed, elapsed, nsteps, warmup, desc, "f32 E-step / f64 M-step")
    out["roofline"] = {"bound": "mfma", "kernel": "k_gemm_nt_f64 (Cholesky of S + triangular solve for diag(Sigma))",
                       "achieved": flops / t_m / 1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": flops / t_m / 1e12 / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                       "algorithmic_flops_per_mstep": flops, "solve_ms": 1e3 * t_m}
    return out


# ----------------------------------------------------------------------------------------------------------------
# C4: FilterReg
# ----------------------------------------------------------------------------------------------------------------
def bench_filterreg(workload, steps, warmup):
    """One step = lattice E-step + Kabsch M-step (filterreg.py:129-146), sigma2 updated every step.  As for C1 the
    window is pinned: the state goes back to the start of the registration after the warm-up steps."""
    import torch
    from probreg_amd import filterreg, math_utils as mu, synthetic

    _kind, n, desc = WORKLOADS[workload]
    src, tgt, (r_true, _) = synthetic.filterreg_pair(n, seed=0)
    reg = filterreg.RigidFilterReg(src, update_sigma2=True)
    plan = reg._ensure_plan(tgt)
    s2_0 = max(mu.squared_kernel_sum(src, tgt), 1e-4)
    state = {}
    sizes = []

    def step():
        size, _blur = plan.estep()                               # one hand-over: the lattice size (device mailbox)
        plan.mstep(0.05, True, "pt2pt", 1e-4, read=False)        # enqueued only; the kernel advances the device state
        sizes.append(size)

    plan.set_state(np.identity(3), np.zeros(3), s2_0)
    for _ in range(warmup):
        step()
    plan.set_state(np.identity(3), np.zeros(3), s2_0)  # back to iteration 0
    del sizes[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    out = plan.get_state()
    state["rot"], state["sigma2"] = out[:9].reshape(3, 3).copy(), out[15]
    d, c = 3, 5
    lat = float(np.mean(sizes))
    alg = (2 * n) * (4 * d + 8 * (d + 1)) + n * (4 * c + 8 * (d + 1) + 8 * c * (d + 1)) \
        + n * (8 * (d + 1) + 4 * c * (d + 1) + 4 * c) + (d + 1) * lat * c * 16
    out = _base("EM iterations/sec (%s)" % desc, steps / elapsed, elapsed, steps, warmup, desc, "f32 lattice / f64 M-step")
    out["config"]["window"] = "EM iterations 0..%d of the registration (state reset after the warm-up steps)" % (steps - 1)
    out["roofline"] = {"bound": "hbm", "kernel": "whole lattice E-step (embed, hash, splat, blur, slice)",
                       "achieved": alg / (elapsed / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg / (elapsed / steps) / 1e9 / HBM_PEAK_GBS,
                       "traffic": _pmc_traffic(workload, "iteration_hbm_bytes"),
                       "algorithmic_bytes_per_iteration": alg, "mean_lattice_vertices": lat,
                       "lattice_vertices_first_last": [sizes[0], sizes[-1]]}
    out["result"] = {"sigma2": state["sigma2"], "rot_err_vs_truth": float(np.max(np.abs(state["rot"] - r_true)))}
    return out


def _slim(line):
    """What an `other_workloads` entry keeps of a workload's own line."""
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "result",
            "dense_it_s", "late_it_s", "kernel_ms", "kernel_factor", "dense_solver")
    out = {k: line[k] for k in keep if k in line}
    rf = out.get("roofline", {})
    for k in ("how", "effective_hbm", "late_regime"):
        rf.pop(k, None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="rigid_100k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline and the parity block")
    ap.add_argument("--no-other-workloads", action="store_true", help="C1 only (skip the C2 / C3 / C4 entries)")
    ap.add_argument("--tuning", default="", help="r_col,seg_col,r_row,seg_row (0 = auto)")
    ap.add_argument("--pairs-log", default="", help="write the per-iteration (engine, pairs, ms) table of the timed window here")
    ap.add_argument("--no-dense-compare", action="store_true",
                    help="non-rigid: skip the M-step through the dense fallback (profiles of the product path alone)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU under torch.distributed.run,
        # exactly the command the docstring gives); rank 0 of that run prints the one JSON line on this stdout
        import socket
        import subprocess

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the two must agree (n_gpus in the line is the "
                         "number of ranks that ran)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); none visible")
    if os.environ.get("PROBREG_SHARE_GPU") == "1":  # test rig only: N ranks on one GPU (use with PROBREG_DIST_BACKEND=gloo)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("PROBREG_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("PROBREG_DIST_BACKEND", "nccl")  # nccl = RCCL; gloo only for the shared-GPU test rig
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    kind, n, desc = WORKLOADS[args.workload]
    if kind in ("nonrigid", "filterreg", "bcpd"):
        if world > 1:
            raise SystemExit("%s runs as single-GPU replicas (DESIGN.md section 6)" % args.workload)
        if kind == "nonrigid":
            out = bench_nonrigid(args.workload, args.steps, args.warmup, dense_compare=not args.no_dense_compare)
        else:
            out = {"filterreg": bench_filterreg, "bcpd": bench_bcpd}[kind](args.workload, args.steps, args.warmup)
        print(json.dumps(out))
        return
    out = bench_cpd(args.workload, args.steps, args.warmup, args.tuning, args.pairs_log or None)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(n)
        if world == 1 and args.workload == "rigid_100k" and not args.no_other_workloads:
            others = {}
            for name, fn, k, w in (("affine_200k", bench_cpd, 20, 1), ("nonrigid_50k", bench_nonrigid, 20, 2),
                                   ("filterreg_500k", bench_filterreg, 20, 3)):
                try:
                    others[name] = _slim(fn(name, k, w))
                except Exception as e:  # a failing side workload must not take the headline line with it
                    others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()
            if not args.no_cpu_baseline:
                try:
                    for name, blk in parity_other_workloads().items():
                        if name in others and "error" not in others[name]:
                            others[name]["parity"] = blk
                except Exception as e:
                    others["parity_error"] = "%s: %s" % (type(e).__name__, e)
            out["other_workloads"] = others
        print(json.dumps(out))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        from probreg_amd import dist as pdist

        pdist.reset_native_comms()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
