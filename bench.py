#!/usr/bin/env python
"""Benchmark of the CPD EM hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one EM iteration (transform + E-step column pass + E-step row pass + fp64 moment
reduction + [all-reduce] + device M-step) of RigidCPD on BASELINE.json's config C1: synthetic
N = M = 100 000 3-D points, fp32 pair arithmetic, w = 0.  With N GPUs the SAME problem is solved
with the target cloud sharded over the ranks ("scaling": "strong"); one all-reduce of 32 doubles
per iteration.  Inputs are resident in HBM before the timed region.  Rank 0 prints one JSON line.
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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak

WORKLOADS = {
    # name: (kind, N = M, description)
    "rigid_100k": ("rigid", 100000, "C1 RigidCPD fp32 synthetic N=M=100000 D=3 w=0"),
    "affine_200k": ("affine", 200000, "C2 AffineCPD fp32 synthetic N=M=200000 D=3 w=0"),
    "nonrigid_50k": ("nonrigid", 50000, "C3 NonRigidCPD fp32 E-step / fp64 M-step synthetic N=M=50000 beta=2 lmd=2"),
    "filterreg_500k": ("filterreg", 500000, "C4 FilterReg rigid pt2pt synthetic N=M=500000, 5% outliers, w=0.05"),
    "bcpd_20k": ("bcpd", 20000, "BCPD (SURVEY 8f rank 4) fp32 E-step / fp64 M-step synthetic N=M=20000 lmd=2 w=0.05"),
    "rigid_20k": ("rigid", 20000, "reduced RigidCPD fp32 synthetic N=M=20000 (debug only)"),
}
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X f64 matrix peak (v_mfma_f64_16x16x4_f64, 32 FLOP/clk/SIMD at 2.4 GHz)


def cpu_baseline(n_full):
    """Oracle timed on the host cores on a bounded sample of the same workload (rank 0, N=1 only)."""
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import synthetic

    ns = 40000
    src, tgt, _ = synthetic.rigid_pair(ns, seed=0)
    sigma2 = co.squared_kernel_sum_closed_form(src, tgt)
    params = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0)
    iters = 2
    t0 = time.perf_counter()
    for _ in range(iters):
        ts = co.transform("rigid", params, src)
        es = co.EstepResult(*cpd_c.expectation_step(ts, tgt, sigma2, 0.0))
        params, sigma2, _q = co.mstep_rigid(src, tgt, es)
    dt = (time.perf_counter() - t0) / iters
    scale = (float(n_full) * n_full) / (float(ns) * ns)
    return {
        "value": 1.0 / (dt * scale),
        "unit": "EM iterations/s",
        "cores": cpd_c.threads(),
        "kind": "port",
        "sample": "oracle/cpd_estep_c.c (C/OpenMP fp64 restatement of probreg cpd.py:71-88) + numpy M-step, "
                  "RigidCPD N=M=%d, %d iterations, %.2f s/iteration measured, scaled by M*N (x%.2f) to N=M=%d"
                  % (ns, iters, dt, scale, n_full),
        "measured_iter_s_at_sample": 1.0 / dt,
    }


def _base(args, metric, value, elapsed, desc, dtype):
    return {"metric": metric, "value": value, "unit": "EM iterations/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc}}


def bench_nonrigid(args, n, desc):
    """C3: one step = E-step (fp32 sweeps) + fp64 Cholesky M-step (cpd.py:284-303)."""
    import torch
    from probreg_amd import cpd, synthetic

    src, tgt = synthetic.nonrigid_pair(n, seed=0)
    reg = cpd.NonRigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan

    def step():
        plan.estep(0.0)
        plan.mstep_nonrigid(2.0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # M-step alone (dominant): HIP-synchronised wall time around prg_cpd_mstep_nonrigid
    plan.estep(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.mstep_nonrigid(2.0)
    torch.cuda.synchronize()
    t_m = time.perf_counter() - t0
    flops = n ** 3 / 3.0 + 3 * 2.0 * n * n * 3   # Cholesky + three G-times-(M x 3) products
    out = _base(args, "EM iterations/sec (%s)" % desc, args.steps / elapsed, elapsed, desc, "f32 E-step / f64 M-step")
    out["roofline"] = {"bound": "mfma", "kernel": "k_gemm_nt_f64 (blocked Cholesky of S = cI + D^1/2 G D^1/2)",
                       "achieved": flops / t_m / 1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": flops / t_m / 1e12 / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                       "algorithmic_flops_per_mstep": flops, "mstep_ms": 1e3 * t_m}
    out["result"] = {"sigma2": float(plan.get_params()[13])}
    return out


def bench_bcpd(args, n, desc):
    """BCPD: one step = weighted E-step + Woodbury M-step (bcpd.py:82-98 loop body without the convergence test)."""
    import torch
    from probreg_amd import bcpd, synthetic

    src, tgt = synthetic.nonrigid_pair(n, seed=0)
    src, tgt = src * 10.0, tgt * 10.0  # object ~20 units across: the c = 1 inverse-multiquadric kernel has a sensible width
    reg = bcpd.CombinedBCPD(src)
    stamps = []
    reg.set_callbacks([lambda tr: (torch.cuda.synchronize(), stamps.append(time.perf_counter()))])
    reg.registration(tgt, w=0.05, maxiter=args.warmup + args.steps, tol=-1.0)
    elapsed = stamps[-1] - stamps[args.warmup - 1] if args.warmup > 0 else stamps[-1] - stamps[0]
    steps = args.steps if args.warmup > 0 else args.steps - 1
    plan = reg._plan
    nu, resid = np.ones(n), np.zeros((n, 3))
    plan.bcpd_solve(2.0, 10.0, resid, nu)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.bcpd_solve(2.0, 10.0, resid, nu)
    torch.cuda.synchronize()
    t_m = time.perf_counter() - t0
    flops = 4.0 / 3.0 * n ** 3  # Cholesky (1/3) + triangular solve with M right-hand sides (1)
    out = _base(args, "EM iterations/sec (%s)" % desc, steps / elapsed, elapsed, desc, "f32 E-step / f64 M-step")
    out["steps"] = steps
    out["ms_per_step"] = 1e3 * elapsed / steps
    out["roofline"] = {"bound": "mfma", "kernel": "k_gemm_nt_f64 (Cholesky of S + triangular solve for diag(Sigma))",
                       "achieved": flops / t_m / 1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": flops / t_m / 1e12 / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                       "algorithmic_flops_per_mstep": flops, "solve_ms": 1e3 * t_m}
    return out


def bench_filterreg(args, n, desc):
    """C4: one step = lattice E-step + Kabsch M-step (filterreg.py:129-146), sigma2 updated every step."""
    import torch
    from probreg_amd import filterreg, math_utils as mu, synthetic

    src, tgt, (r_true, _) = synthetic.filterreg_pair(n, seed=0)
    reg = filterreg.RigidFilterReg(src, update_sigma2=True)
    plan = reg._ensure_plan(tgt)
    state = {"rot": np.identity(3), "t": np.zeros(3), "sigma2": max(mu.squared_kernel_sum(src, tgt), 1e-4)}
    sizes = []

    plan.set_state(state["rot"], state["t"], state["sigma2"])  # once: the M-step kernel advances the device state

    def step():
        size, _blur = plan.estep()
        out = plan.mstep(0.05, True, "pt2pt", 1e-4)
        state["rot"], state["t"], state["sigma2"] = out[:9].reshape(3, 3).copy(), out[9:12].copy(), out[12]
        sizes.append(size)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    d, c = 3, 5
    lat = float(np.mean(sizes[args.warmup:]))
    alg = (2 * n) * (4 * d + 8 * (d + 1)) + n * (4 * c + 8 * (d + 1) + 8 * c * (d + 1)) \
        + n * (8 * (d + 1) + 4 * c * (d + 1) + 4 * c) + (d + 1) * lat * c * 16
    out = _base(args, "EM iterations/sec (%s)" % desc, args.steps / elapsed, elapsed, desc, "f32 lattice / f64 M-step")
    out["roofline"] = {"bound": "hbm", "kernel": "whole lattice E-step (embed, hash, splat, blur, slice)",
                       "achieved": alg / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "algorithmic_bytes_per_iteration": alg, "mean_lattice_vertices": lat}
    out["result"] = {"sigma2": state["sigma2"], "rot_err_vs_truth": float(np.max(np.abs(state["rot"] - r_true)))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="rigid_100k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tuning", default="", help="r_col,seg_col,r_row,seg_row (0 = auto)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    from probreg_amd import _lib, cpd, synthetic

    kind, n, desc = WORKLOADS[args.workload]
    if kind in ("nonrigid", "filterreg", "bcpd"):
        if world > 1:
            raise SystemExit("%s runs as single-GPU replicas (DESIGN.md section 6)" % args.workload)
        out = {"nonrigid": bench_nonrigid, "filterreg": bench_filterreg, "bcpd": bench_bcpd}[kind](args, n, desc)
        print(json.dumps(out))
        return
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
    if args.tuning:
        plan.set_tuning(*[int(v) for v in args.tuning.split(",")])

    def step():
        plan.estep(0.0)
        reg._all_reduce_moments(plan)
        plan.mstep(kind_id, True)

    def fence():
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    start_params = plan.get_params()  # EM state at the start of the timed region (replayed below)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    res = reg._result_from_params(plan.get_params())

    # Per-kernel timing with HIP events on the plan's stream: the SAME K iterations replayed from the saved
    # state (the sweeps skip provably-zero blocks, so their duration depends on sigma2 and must be averaged
    # over the same trajectory the timed region walked).
    plan.set_params(start_params)
    acc, first = {}, None
    for _ in range(args.steps):
        ms = plan.estep_timed(0.0)
        reg._all_reduce_moments(plan)
        plan.mstep(kind_id, True)
        first = first or dict(ms)
        for k, v in ms.items():
            acc[k] = acc.get(k, 0.0) + v / args.steps
    last = dict(ms)

    if rank == 0:
        m_pts, n_loc = plan.m, plan.n
        # algorithmic bytes of the reference's formulation at fp32: P (M x N fp32) is written
        # once by the column pass and read once by the row pass; + the clouds and the per-point outputs.
        row_bytes = 4.0 * m_pts * n_loc + 4.0 * (m_pts + n_loc) * 5
        col_bytes = 4.0 * m_pts * n_loc + 4.0 * (m_pts + n_loc) * 5
        row_s, col_s = acc["rowpass"] * 1e-3, acc["colpass"] * 1e-3
        achieved = row_bytes / row_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get("rowpass_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "EM iterations/sec (RigidCPD, N=M=100k fp32)" if args.workload == "rigid_100k"
                      else "EM iterations/sec (%s)" % desc,
            "value": args.steps / elapsed,
            "unit": "EM iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "target_sharding": "contiguous runs of the target's Morton order over %d rank(s)" % world,
                       "collective": "1 all-reduce of 32 fp64 per iteration" if world > 1 else "none",
                       "m": m_pts, "n_local": n_loc, "n_global": n},
            "roofline": {
                "bound": "hbm",
                "kernel": "k_rowpass (E-step sweep 2: P1, PX, sigma2 residual)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": row_bytes,
                "avg_launch_ms": acc["rowpass"],
                "colpass": {"achieved": col_bytes / col_s / 1e9, "frac": col_bytes / col_s / 1e9 / HBM_PEAK_GBS,
                            "avg_launch_ms": acc["colpass"]},
                "e_step": {"algorithmic_bytes": row_bytes + col_bytes, "ms": acc["total"],
                           "frac": (row_bytes + col_bytes) / (acc["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "dense_regime": {"avg_launch_ms": first["rowpass"],
                                 "frac": row_bytes / (first["rowpass"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "what": "first timed iteration: sigma2 still large, no (wave, group) block is "
                                         "skipped - the sweep's VALU-bound throughput on all M x N pairs"},
                "note": "algorithmic bytes = the reference formulation's irreducible fp32 traffic (8 B per "
                        "source-target pair per E-step); the fused kernels keep P in registers and skip blocks "
                        "whose every pair is an exact fp32 zero, so physical HBM traffic is MBs and the kernels "
                        "are VALU/transcendental bound (DESIGN.md section 3.1)",
            },
            "kernel_ms": {"mean_over_timed_iterations": acc, "first_timed_iteration": first,
                          "last_timed_iteration": last},
            "result": {"sigma2": res.sigma2, "q": res.q},
        }
        if kind == "rigid":
            r_true, t_true, _ = truth
            out["result"]["rot_err_vs_truth"] = float(np.max(np.abs(res.transformation.rot - r_true)))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(out))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
