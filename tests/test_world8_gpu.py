"""SURVEY.md 8(e) with the world size BASELINE.json's metric is quoted on: EIGHT ranks, one process each, the target cut into
eight spatial shards (probreg_amd.dist.spatial_shard), the real sharded kernels on every rank and the per-iteration all-reduce of
the moment block between them.  The test box has one GPU, so the eight ranks share cuda:0 and the collective is gloo's (RCCL
refuses several ranks on one device); launcher, shard cut, engine decisions per rank (each rank leaves the matrix cores at its OWN
iteration - the decision goes by local pair counts), MOMENTS block, device M-step are the code an 8-GPU node runs.

  test_eight_ranks_...            the product's registration loop on 8 shards of >= 8192 columns each (the matrix-core engines are
                                  open to every rank) against the UNSHARDED fp64 oracle at the north-star tolerances
  test_bench_gpus_8_...           `python bench.py --gpus 8` on C1 itself: the launcher starts 8 ranks, the line says n_gpus = 8 and
                                  the EM state after the window is the 1-rank run's
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_SIGMA2 = 1e-5
TOL_TF = 1e-4
WORLD = 8
N8, M8, K8 = 66000, 24000, 18   # 8250 columns per rank (>= 8192: fused matrix-core sweep available), 18 iterations, w = 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from probreg_amd import cpd, dist, synthetic

        src, tgt, _ = synthetic.rigid_pair(N8, m=M8, seed=61)
        reg = cpd.RigidCPD(src)
        reg._initialize(tgt)
        plan = reg._plan
        assert plan.n == len(dist.spatial_shard(tgt, rank, world))
        plan.set_moments_only(1)   # as RigidCPD.registration does for its own loop
        single, engine = [], []
        for _ in range(K8):
            plan.estep(0.0)
            single.append(plan.last_estep_fused())
            engine.append(plan.last_estep_engine())
            reg._all_reduce_moments(plan)
            reg._device_mstep(plan)
        res = reg._result_from_params(plan.get_params())
        ret[rank] = dict(sigma2=float(res.sigma2), q=float(res.q), rot=np.array(res.transformation.rot), t=np.array(res.transformation.t),
                         scale=float(res.transformation.scale), single=single, engine=engine, n_local=int(plan.n))
    finally:
        tdist.destroy_process_group()


def test_eight_ranks_on_one_gpu_match_the_unsharded_oracle():
    import torch.multiprocessing as mp

    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import synthetic

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(WORLD, _free_port(), ret), nprocs=WORLD, join=True)
    out = [ret[r] for r in range(WORLD)]
    assert sum(o["n_local"] for o in out) == N8 and max(o["n_local"] for o in out) - min(o["n_local"] for o in out) <= 1
    for o in out:
        assert o["single"] == [1] * K8          # one pair sweep per rigid iteration on every rank, in every regime
        assert o["engine"][0] == 1 and o["engine"] == sorted(o["engine"], reverse=True), o["engine"]  # hand over once, for good
        # every rank ends in the same state: same all-reduced numbers, same deterministic M-step
        assert o["sigma2"] == out[0]["sigma2"] and np.array_equal(o["rot"], out[0]["rot"]) and np.array_equal(o["t"], out[0]["t"])
    handover = [o["engine"].index(0) if 0 in o["engine"] else K8 for o in out]
    print("8-rank shards: each rank's first vector-pipe iteration %s" % handover)
    src, tgt, _ = synthetic.rigid_pair(N8, m=M8, seed=61)
    params, sigma2 = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0), co.squared_kernel_sum_closed_form(src, tgt)
    for _ in range(K8):
        es = co.EstepResult(*cpd_c.expectation_step(co.transform("rigid", params, src), tgt, sigma2, 0.0))
        params, sigma2, _q = co.mstep_rigid(src, tgt, es, update_scale=True)
    a = out[0]
    print("8-rank shards vs unsharded oracle: sigma2 %.2e rot %.2e t %.2e" % (
        abs(a["sigma2"] - sigma2) / sigma2, np.max(np.abs(a["rot"] - params["rot"])), np.max(np.abs(a["t"] - params["t"]))))
    assert abs(a["sigma2"] - sigma2) <= TOL_SIGMA2 * sigma2, (a["sigma2"], sigma2)
    assert np.max(np.abs(a["rot"] - params["rot"])) <= TOL_TF
    assert np.max(np.abs(a["t"] - params["t"])) <= TOL_TF
    assert abs(a["scale"] - params["scale"]) <= TOL_TF * params["scale"]


def _bench_line(args, extra_env, timeout=1200):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=timeout)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_8_runs_c1_on_eight_ranks():
    """The driver's scaling command at N = 8 (`python bench.py --gpus 8`, C1 = rigid_100k, K = 20) on the shared-GPU rig: eight
    ranks started by the launcher, 12 500 columns each, and the EM state after the window equal to the one-rank run's."""
    common = ["--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-other-workloads"]
    one = _bench_line(["--gpus", "1"] + common, {})
    eight = _bench_line(["--gpus", "8"] + common, {"PROBREG_SHARE_GPU": "1", "PROBREG_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and eight["n_gpus"] == 8
    assert eight["config"]["n_global"] == 100000 and eight["config"]["n_local"] == 12500
    assert eight["scaling"] == "strong" and "gloo" in eight["config"]["collective"]
    assert abs(one["result"]["sigma2"] - eight["result"]["sigma2"]) <= TOL_SIGMA2 * one["result"]["sigma2"]
    assert abs(one["result"]["q"] - eight["result"]["q"]) <= TOL_SIGMA2 * abs(one["result"]["q"])
