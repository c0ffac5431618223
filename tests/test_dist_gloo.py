"""The N > 1 path on CPU: two processes, torch.distributed gloo backend.

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
