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
            err_nr = np.max(np.abs(g @ a["w"] - disp)) / max(1.0, np.max(np.abs(disp)))
            print("sharded non-rigid displacement error %.2e" % err_nr)
            assert err_nr <= TOL_T  # (round 3 granted 3e-4 here; measured 3.3e-7)
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


def _bench_line(args, extra_env, timeout=900):
    import json
    import subprocess
    import sys

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


_COMMON = ["--steps", "6", "--warmup", "1", "--workload", "rigid_20k", "--no-cpu-baseline", "--no-other-workloads"]
_PLAIN = {}


def _plain_line():
    """The one-rank line of the reduced workload, run ONCE for the launcher tests below (every `python bench.py` is a fresh
    process that pays for `import torch` - on a cold box a minute each)."""
    if "line" not in _PLAIN:
        _PLAIN["line"] = _bench_line(["--gpus", "1"] + _COMMON, {})
    return _PLAIN["line"]


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks itself and report n_gpus = 2
    (round 3: --gpus was parsed and ignored - an 8-GPU scaling run would have recorded one rank).  One GPU here, so both
    ranks share it and the collective is gloo's; launcher, sharding, per-iteration all-reduce and the line are the code an
    8-GPU node runs.  The EM state after the window must be the 1-rank run's."""
    one = _plain_line()
    two = _bench_line(["--gpus", "2"] + _COMMON, {"PROBREG_SHARE_GPU": "1", "PROBREG_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["n_local"] * 2 == two["config"]["n_global"]
    assert "gloo" in two["config"]["collective"]
    assert abs(one["result"]["sigma2"] - two["result"]["sigma2"]) <= 1e-5 * one["result"]["sigma2"]
    assert abs(one["result"]["q"] - two["result"]["q"]) <= 1e-5 * abs(one["result"]["q"])


def test_world_size_and_gpus_flag_must_agree():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--workload", "rigid_20k"], cwd=root,
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert run.returncode != 0 and "must agree" in (run.stderr + run.stdout)


def test_library_side_rccl_all_reduce_one_rank():
    """The per-iteration collective inside the C-ABI (prg_comm_*, ncclAllReduce on the plan's stream): with a one-rank
    communicator the registration must be BIT-identical to the plain one, and the library must have issued exactly one
    all-reduce per EM iteration plus one for the sigma2 initialiser (+ the probe of the communicator's creation)."""
    import torch

    from probreg_amd import cpd, dist, synthetic

    src, tgt, _ = synthetic.rigid_pair(20000, seed=5)
    plain = cpd.registration_cpd(src, tgt, "rigid", maxiter=12, tol=-1.0)
    os.environ["PROBREG_NATIVE_RCCL"] = "1"
    try:
        dist.reset_native_comms()
        comm = dist.native_comm(torch.cuda.current_device())
        assert comm is not None and comm.nranks == 1
        before = comm.calls()
        reg = cpd.RigidCPD(src)
        res = reg.registration(tgt, maxiter=12, tol=-1.0)
        assert reg._plan._comm is comm
        assert comm.calls() - before == 12 + 1
        assert res.sigma2 == plain.sigma2 and res.q == plain.q
        assert np.array_equal(res.transformation.rot, plain.transformation.rot)
        # with a tolerance to test (host reads q every iteration) the Python loop runs: same collective, same numbers
        res2 = cpd.registration_cpd(src, tgt, "rigid", maxiter=12, tol=0.0)
        assert res2.sigma2 == plain.sigma2
        # a reused plan must not keep a communicator that is gone: reset_native_comms detaches it before destroying it, and
        # the next registration of the same object (no communicator wanted any more) runs plain
        dist.reset_native_comms()
        assert reg._plan._comm is None
        os.environ["PROBREG_NATIVE_RCCL"] = "0"
        res3 = reg.registration(tgt, maxiter=12, tol=-1.0)
        assert res3.sigma2 == plain.sigma2 and np.array_equal(res3.transformation.rot, plain.transformation.rot)
        os.environ["PROBREG_NATIVE_RCCL"] = "1"
        # non-rigid: the per-point block goes through the library's all-reduce as well
        s_n, t_n = synthetic.nonrigid_pair(3000, seed=6)
        os.environ["PROBREG_NATIVE_RCCL"] = "0"
        dist.reset_native_comms()
        ref = cpd.registration_cpd(s_n, t_n, "nonrigid", maxiter=4, tol=-1.0)
        os.environ["PROBREG_NATIVE_RCCL"] = "1"
        dist.reset_native_comms()
        got = cpd.registration_cpd(s_n, t_n, "nonrigid", maxiter=4, tol=-1.0)
        assert got.sigma2 == ref.sigma2 and np.array_equal(got.transformation.w, ref.transformation.w)
    finally:
        os.environ.pop("PROBREG_NATIVE_RCCL", None)
        dist.reset_native_comms()


def test_nccl_process_group_selects_the_library_side_collective():
    """A (single-rank) nccl process group: `bench.py` must report the library-side RCCL collective in its line - the path an
    8-GPU run takes - and reproduce the run without any process group."""
    plain = _plain_line()
    forced = _bench_line(["--gpus", "1"] + _COMMON, {"PROBREG_FORCE_DIST": "1", "MASTER_PORT": str(_free_port())})
    assert forced["config"]["collective_path"].startswith("library-side RCCL"), forced["config"]
    assert forced["result"]["sigma2"] == plain["result"]["sigma2"] and forced["result"]["q"] == plain["result"]["q"]
