"""SURVEY.md 8(e) on the real kernels: TWO ranks, one process each, target rows sharded, the HIP E-step on every
rank's shard and the per-iteration all-reduce of the moment block (rigid / affine) or of the per-point block
(non-rigid) between them.  The test box has one GPU, so both ranks use cuda:0 and the collective goes through the
``gloo`` backend (RCCL refuses two ranks on one device); everything else - shard bounds, Morton sort of the local
rows, the bound moment tensor, the in-place all-reduce on the plan's stream, the device M-step - is the code that
runs under ``torchrun`` with ``nccl`` on an 8-GPU node.  Checked against the oracle on the unsharded problem."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_T = 1e-4
TOL_SIGMA2 = 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cases():
    from probreg_amd import synthetic

    src_r, tgt_r, _ = synthetic.rigid_pair(5003, m=4100, seed=81)  # odd N: the two shards differ in size
    src_a, tgt_a, _ = synthetic.affine_pair(3001, m=2600, seed=82)
    src_n, tgt_n = synthetic.nonrigid_pair(1101, m=900, seed=83)
    return {"rigid": (src_r, tgt_r, dict(w=0.1, maxiter=8)), "affine": (src_a, tgt_a, dict(w=0.05, maxiter=6)),
            "nonrigid": (src_n, tgt_n, dict(w=0.0, maxiter=3))}


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from probreg_amd import cpd, dist

        assert dist.world() == (rank, world)
        for kind, (src, tgt, kw) in _cases().items():
            reg_res = cpd.registration_cpd(src, tgt, kind, tol=-1.0, **kw)
            tf = reg_res.transformation
            out = {"sigma2": float(reg_res.sigma2), "q": float(reg_res.q)}
            if kind == "rigid":
                out.update(lin=np.array(tf.rot), t=np.array(tf.t), scale=float(tf.scale))
            elif kind == "affine":
                out.update(lin=np.array(tf.b), t=np.array(tf.t))
            else:
                out.update(w=np.array(tf.w))
            ret["%s_%d" % (kind, rank)] = out
    finally:
        tdist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_unsharded_oracle():
    import torch.multiprocessing as mp

    from oracle import cpd_numpy as co

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for kind, (src, tgt, kw) in _cases().items():
        a, b = ret["%s_0" % kind], ret["%s_1" % kind]
        p, s2, q, _ = co.registration(kind, src, tgt, tol=-1.0, closed_form_init=True, **kw)
        # every rank ends with the same state (same all-reduced numbers, same deterministic M-step)
        assert a["sigma2"] == b["sigma2"] and a["q"] == b["q"], kind
        assert abs(a["sigma2"] - s2) <= TOL_SIGMA2 * s2, (kind, a["sigma2"], s2)
        if kind == "nonrigid":
            assert np.array_equal(a["w"], b["w"])
            g = co.rbf_kernel(src, src, 2.0).astype(np.float64)  # compare displacements G W (W itself is ill-conditioned)
            disp = g @ p["w"]
            assert np.max(np.abs(g @ a["w"] - disp)) <= 3e-4 * max(1.0, np.max(np.abs(disp)))
        else:
            assert np.array_equal(a["lin"], b["lin"]) and np.array_equal(a["t"], b["t"])
            lin = p["rot"] if kind == "rigid" else p["b"]
            assert np.max(np.abs(a["lin"] - lin)) <= TOL_T, kind
            assert np.max(np.abs(a["t"] - p["t"])) <= TOL_T * max(1.0, np.max(np.abs(p["t"]))), kind
            if kind == "rigid":
                assert abs(a["scale"] - p["scale"]) <= TOL_T * p["scale"]


def test_bench_under_torchrun_with_rccl_when_two_gpus_are_present():
    """RCCL with real peers: `torchrun --nproc-per-node 2 bench.py --gpus 2` (one rank per GPU, backend nccl = RCCL,
    the per-iteration all-reduce of the 32-double moment block over xGMI) must reproduce the single-GPU EM state.
    Self-skips on a one-GPU box (the driver's multi-GPU node is the first place this can run)."""
    import json
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "6", "--warmup", "1", "--workload", "rigid_20k", "--no-cpu-baseline", "--no-other-workloads"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, cwd=root, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(root, "bench.py"), "--gpus", "2"] + common, cwd=root, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["value"] > 0
    assert abs(a["result"]["sigma2"] - b["result"]["sigma2"]) <= 1e-5 * a["result"]["sigma2"]
    assert abs(a["result"]["q"] - b["result"]["q"]) <= 1e-5 * abs(a["result"]["q"])
