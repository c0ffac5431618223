"""The N > 1 path on CPU: two processes, torch.distributed gloo backend.

Two layers: (1) ``test_product_driver_*`` run the PRODUCT's own host code under two ranks - ``probreg_amd.cpd``'s
registration driver, ``dist.spatial_shard``, the fp64 centring and the in-place all-reduce of the bound moment tensor
- with ``tests/oracle_plan.py`` standing in for the GPU plan (there is no GPU here); (2) the older arithmetic-only
check of the shard-additive moment block.

What runs here is the host logic of the sharded EM iteration: every rank takes its contiguous
block of target rows (probreg_amd.dist.shard_bounds), produces the 32-double MOMENTS block for its shard,
one in-place SUM all-reduce (probreg_amd.dist.all_reduce_sum_) combines them, and every rank runs the same
M-step.  On a GPU box the shard moments come from the HIP kernels; without a GPU the oracle's E-step stands
in for them - the point of this test is the sharding arithmetic and the collective plumbing, which are
identical in both cases (tests/test_cpd_gpu.py::test_shards_sum_to_whole_on_one_gpu covers the kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from oracle import cpd_numpy as co
from probreg_amd import dist, synthetic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert dist.world() == (rank, world)
        if kind == "rigid":
            src, tgt, _ = synthetic.rigid_pair(900, m=700, seed=3)
        else:
            src, tgt, _ = synthetic.affine_pair(900, m=700, seed=4)
        lo, hi = dist.shard_bounds(tgt.shape[0], rank, world)
        # sigma2 initialiser: local target sums, all-reduced (MOMENTS[24..27])
        t_local = tgt[lo:hi]
        sums = torch.zeros(32, dtype=torch.float64)
        sums[24:27] = torch.from_numpy(t_local.sum(axis=0))
        sums[27] = float(np.sum(t_local * t_local))
        dist.all_reduce_sum_(sums)
        m, n, d = src.shape[0], tgt.shape[0], 3
        sigma2 = (m * sums[27].item() + n * np.sum(src * src)
                  - 2.0 * np.dot(sums[24:27].numpy(), src.sum(axis=0))) / (d * m * n)
        params = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0) if kind == "rigid" else dict(b=np.identity(3),
                                                                                                 t=np.zeros(3))
        w = 0.1
        for _ in range(3):
            ts = co.transform(kind, params, src)
            # local E-step against the GLOBAL denominator: den needs all source points but only the local
            # target columns, so a column shard is self-contained; `c` uses the global N (cpd.py:78-79)
            inv = -1.0 / (2.0 * sigma2)
            dd = ((ts[:, None, :] - t_local[None, :, :]) ** 2).sum(axis=2)
            k = np.exp(dd * inv)
            c = (2.0 * np.pi * sigma2) ** 1.5 * w / (1.0 - w) * m / n
            den = k.sum(axis=0)
            den[den == 0] = co.EPS32
            den += c
            p = k / den
            es = co.EstepResult(p.sum(axis=0), p.sum(axis=1), p @ t_local, float(p.sum()))
            mom = torch.from_numpy(co.moments_from_estep(src, t_local, es))
            dist.all_reduce_sum_(mom)          # THE collective: 32 doubles per iteration
            res = co.mstep_from_moments(kind, mom.numpy(), 3)
            params, sigma2 = res.params, res.sigma2
        # the numpy helper goes through the same collective
        tot = dist.all_reduce_sum_numpy(np.array([float(hi - lo)]))
        assert int(tot[0]) == n
        if rank == 0:
            ret["sigma2"] = float(sigma2)
            ret["lin"] = np.asarray(params["rot"] if kind == "rigid" else params["b"]).copy()
            ret["t"] = np.asarray(params["t"]).copy()
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("kind", ["rigid", "affine"])
def test_two_rank_sharded_em_matches_single_process(kind):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, kind, ret), nprocs=world, join=True)
    if kind == "rigid":
        src, tgt, _ = synthetic.rigid_pair(900, m=700, seed=3)
    else:
        src, tgt, _ = synthetic.affine_pair(900, m=700, seed=4)
    p, s2, q, _ = co.registration(kind, src, tgt, w=0.1, maxiter=3, tol=-1.0, closed_form_init=True)
    assert abs(ret["sigma2"] - s2) < 1e-10 * s2
    assert np.max(np.abs(ret["lin"] - (p["rot"] if kind == "rigid" else p["b"]))) < 1e-9
    assert np.max(np.abs(ret["t"] - p["t"])) < 1e-9


# ---------------------------------------------------------------------------------------------------------------
# the product's driver under two ranks (oracle-backed plan instead of the GPU one)
# ---------------------------------------------------------------------------------------------------------------
def _driver_cases():
    src_r, tgt_r, _ = synthetic.rigid_pair(701, m=600, seed=13)  # odd N: shards of different sizes
    src_a, tgt_a, _ = synthetic.affine_pair(640, m=500, seed=14)
    return {"rigid": (src_r + 3.0, tgt_r - 2.0, dict(w=0.1, maxiter=5)),   # offsets: the centring has work to do
            "rigid_noscale": (src_r, tgt_r, dict(w=0.0, maxiter=4, update_scale=False)),
            "affine": (src_a, tgt_a, dict(w=0.05, maxiter=4))}


def _driver_worker(rank, world, port, ret):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_plan import OraclePlan
    from probreg_amd import cpd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cpd.CpdPlan = OraclePlan  # the only substitution: everything above the C ABI is the product's code
        for name, (src, tgt, kw) in _driver_cases().items():
            kind = "affine" if name == "affine" else "rigid"
            seen = []
            res = cpd.registration_cpd(src, tgt, kind, tol=-1.0, callbacks=[lambda t: seen.append(1)], **kw)
            tf_ = res.transformation
            ret["%s_%d" % (name, rank)] = dict(lin=np.array(tf_.rot if kind == "rigid" else tf_.b), t=np.array(tf_.t),
                                               sigma2=float(res.sigma2), q=float(res.q), ncb=len(seen),
                                               n_local=int(cpd_plan_rows(src, tgt, rank, world)))
    finally:
        tdist.destroy_process_group()


def cpd_plan_rows(src, tgt, rank, world):
    return len(dist.spatial_shard(tgt, rank, world))


def test_product_driver_two_ranks_matches_unsharded_oracle():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_driver_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for name, (src, tgt, kw) in _driver_cases().items():
        kind = "affine" if name == "affine" else "rigid"
        a, b = ret["%s_0" % name], ret["%s_1" % name]
        assert a["n_local"] + b["n_local"] == tgt.shape[0] and abs(a["n_local"] - b["n_local"]) <= 1
        assert a["ncb"] == kw["maxiter"]
        # every rank ends in the same state: same all-reduced numbers, same deterministic M-step
        assert a["sigma2"] == b["sigma2"] and np.array_equal(a["lin"], b["lin"]) and np.array_equal(a["t"], b["t"])
        p, s2, q, _ = co.registration(kind, src, tgt, tol=-1.0, closed_form_init=True, **kw)
        # (the plan keeps the centred clouds as float32, as the GPU one does: agreement at float32-input level)
        assert abs(a["sigma2"] - s2) < 1e-6 * s2, name
        assert abs(a["q"] - q) < 1e-5 * abs(q), name
        assert np.max(np.abs(a["lin"] - (p["rot"] if kind == "rigid" else p["b"]))) < 1e-6, name
        assert np.max(np.abs(a["t"] - p["t"])) < 1e-6, name


def test_spatial_shards_partition_the_target():
    """Morton-order shards: disjoint, complete, near-equal, identical on every rank; world 1 keeps the caller's order."""
    tgt = synthetic.surface(1003, seed=5)
    for world in (1, 2, 3, 8):
        rows = [dist.spatial_shard(tgt, r, world) for r in range(world)]
        allr = np.concatenate(rows)
        assert np.array_equal(np.sort(allr), np.arange(1003))
        assert max(len(r) for r in rows) - min(len(r) for r in rows) <= 1
    assert np.array_equal(dist.spatial_shard(tgt, 0, 1), np.arange(1003))
    with pytest.raises(ValueError):
        dist.shard_bounds(10, 3, 3)
