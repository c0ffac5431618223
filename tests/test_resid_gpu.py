"""The residual-form single sweep of a rigid EM iteration on the VECTOR pipe (csrc/cpd_sweeps_packed.hip
k_colpass_cull<true>, csrc/cpd_sweeps_queue.hip k_colpass_queue<true>, csrc/cpd.hip k_colfinal_resid / k_fused_final): below the
dense regime the rigid M-step's moments (cpd.py:160-192) come from per-column sums A = sum K, U = sum K (x - z), R = sum K |x - z|^2
of ONE culled sweep instead of a column pass and a row pass.  Held to the two-sweep engine from the same state (grid of culled
waves and work queue), to the fp64 oracle along whole registrations into the deep sparse regime (north-star tolerances), on
2-rank shards whose ranks hand over from the matrix cores at their own iteration, and on ranks that disagree on the engine."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_SIGMA2 = 1e-5
TOL_TF = 1e-4


def _oracle_iterations(src, tgt, params, sigma2, k, w=0.0, update_scale=True):
    from oracle import cpd_c, cpd_numpy as co

    q = None
    for _ in range(k):
        es = co.EstepResult(*cpd_c.expectation_step(co.transform("rigid", params, src), tgt, sigma2, w))
        params, sigma2, q = co.mstep_rigid(src, tgt, es, update_scale=update_scale)
    return params, sigma2, q


@pytest.mark.parametrize("n,m,w,sparse,warm", [(40000, 40000, 0.0, 2, 14), (30011, 45007, 0.1, 0, 12), (9000, 12000, 0.0, 1, 3),
                                               (40000, 36000, 0.05, 1, 22)])
def test_resid_sweep_moments_equal_the_two_sweep_engine(n, m, w, sparse, warm):
    """From the same state (after `warm` EM iterations: culling at work, the 40k cases in the sparse regime) the single sweep's
    moments, pt1 and M-step against the column pass + row pass of the vector pipe; sparse 2 = the work queue, 0 = the grid of
    culled waves, 1 = the plan's own choice."""
    from probreg_amd import _lib, cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=41)
    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_dense_engine(0)
    plan.set_sparse_engine(sparse)
    plan.set_moments_only(2)
    for _ in range(warm):
        plan.estep(w)
        plan.mstep(_lib.PRG_TF_RIGID, True)
    state = plan.get_params()
    out = {}
    for mode in (2, 1):
        plan.set_moments_only(mode)
        plan.set_params(state)
        plan.estep(w)
        plan.set_params(state)
        plan.estep(w)    # (the second one culls with the seeds of the first: zero motion)
        assert plan.last_estep_fused() == (1 if mode == 1 else 0)
        assert plan.last_estep_engines() == (0, 0) and plan.last_estep_lean() == 0
        mom = plan.get_moments()
        pt1 = plan.get_estep_pt1()
        col_pairs, row_pairs = plan.pair_counts()
        assert col_pairs > 0 and (row_pairs == 0) == (mode == 1)
        plan.mstep(_lib.PRG_TF_RIGID, True)
        out[mode] = (mom, pt1, plan.get_params(), col_pairs)
    a, b = out[2][0], out[1][0]
    n_p = a[0]
    assert abs(a[0] - b[0]) < 2e-6 * n_p
    assert np.max(np.abs(a[1:16] - b[1:16])) < 2e-6 * n_p           # Sx, Sy, Sxy
    assert abs((a[16] + a[19] + a[21]) - b[16]) < 2e-6 * n_p and np.all(b[17:22] == 0.0)   # tr Syy
    assert abs(a[22] - b[22]) < 2e-6 * n_p                          # sum pt1 |x|^2
    assert np.max(np.abs(out[2][1] - out[1][1])) < 1e-6             # pt1
    if sparse == 1:
        # [r6] the default single sweep is the owner sweep (csrc/cpd_sweeps_owner.hip): the same group boxes and bound, but a wave owns
        # 64 columns instead of 128 (a smaller box: measured 27.6e6 pairs against 39.1e6 here) and the last column block's pad-only
        # groups stay out of its box - never more pairs than the two-sweep column pass
        assert 0.4 * out[2][3] <= out[1][3] <= out[2][3], (out[2][3], out[1][3])
    else:
        assert out[2][3] == out[1][3]                               # the same blocks of pairs evaluated
    pa, pb = out[2][2], out[1][2]
    assert np.max(np.abs(pa[:13] - pb[:13])) < 2e-6
    assert abs(pa[13] - pb[13]) <= 3e-6 * pa[13]
    # no per-point arrays after a single-sweep E-step, and no affine M-step from its moments: the interface says so
    plan.set_params(state)
    plan.estep(w)
    with pytest.raises(_lib.ProbregHipError, match="single sweep"):
        plan.get_estep()
    with pytest.raises(_lib.ProbregHipError, match="affine"):
        plan.mstep(_lib.PRG_TF_AFFINE, True)
    # prg_cpd_set_resid_sweep(0): the vector pipe keeps its two sweeps
    plan.set_resid_sweep(False)
    plan.set_params(state)
    plan.estep(w)
    assert plan.last_estep_fused() == 0
    plan.get_estep()


@pytest.mark.parametrize("case", ["w01_20k", "two_d", "scale_fixed", "queue_24k", "vector_only"])
def test_registration_through_the_single_sweeps_matches_the_oracle(case):
    """`registration` with tol < 0 runs prg_cpd_iterate: every rigid E-step is ONE sweep - the fused matrix-core sweep while
    sigma2 is large, the residual-form vector sweep afterwards - for 30+ iterations, into the sparse regime.  Against the
    reference's loop (C E-step + numpy M-step, fp64)."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    kw, setup = dict(), None
    if case == "w01_20k":
        src, tgt, _ = synthetic.rigid_pair(20000, m=18000, seed=43)
        k, w = 32, 0.1
    elif case == "two_d":
        src, tgt, _ = synthetic.rigid_pair(14000, m=16000, seed=45)
        src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
        k, w = 30, 0.0
    elif case == "scale_fixed":
        src, tgt, _ = synthetic.rigid_pair(16000, seed=47)
        k, w, kw = 30, 0.2, dict(update_scale=False)
    elif case == "queue_24k":
        src, tgt, _ = synthetic.rigid_pair(24000, seed=49)
        k, w = 30, 0.0
        setup = lambda plan: plan.set_sparse_engine(2)   # noqa: E731
    else:
        src, tgt, _ = synthetic.rigid_pair(12000, m=15000, seed=51)
        k, w = 28, 0.05
        setup = lambda plan: plan.set_dense_engine(0)    # noqa: E731  (the residual-form sweep from the first iteration on)
    reg = cpd.RigidCPD(src, **kw)
    if setup is not None:
        reg._initialize(tgt)
        setup(reg._plan)   # (engine modes are plan state: they survive the second upload of `registration`)
    res = reg.registration(tgt, w=w, maxiter=k, tol=-1.0)
    assert reg._plan.last_estep_fused() == 1 and reg._plan.last_estep_engines()[1] == 0
    if case != "scale_fixed":   # (with the scale pinned at 1 sigma2 settles above the noise level: that run may end on the matrix cores)
        assert reg._plan.last_estep_engine() == 0
    dim = src.shape[1]
    s2_0 = co.squared_kernel_sum_closed_form(src, tgt)
    p, s2, q = _oracle_iterations(src, tgt, dict(rot=np.identity(dim), t=np.zeros(dim), scale=1.0), s2_0, k, w,
                                  kw.get("update_scale", True))
    tr = res.transformation
    assert np.max(np.abs(tr.rot - p["rot"])) < TOL_TF
    assert np.max(np.abs(tr.t - p["t"])) < TOL_TF * max(1.0, np.max(np.abs(p["t"])))
    assert abs(tr.scale - p["scale"]) < TOL_TF * p["scale"]
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2, (res.sigma2, s2)
    assert abs(res.q - q) <= 1e-4 * abs(q)


_ORACLE_CACHE = {}


def _shard_oracle():
    from oracle import cpd_numpy as co
    from probreg_amd import synthetic

    if "r" not in _ORACLE_CACHE:
        src, tgt, _ = synthetic.rigid_pair(N_SHARD, seed=53)
        _ORACLE_CACHE["r"] = _oracle_iterations(src, tgt, dict(rot=np.identity(3), t=np.zeros(3), scale=1.0),
                                                co.squared_kernel_sum_closed_form(src, tgt), K_SHARD, W_SHARD)
    return _ORACLE_CACHE["r"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_SHARD, K_SHARD, W_SHARD = 24000, 26, 0.0


def _shard_worker(rank, world, port, ret, modes):
    import torch
    import torch.distributed as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from probreg_amd import cpd, synthetic

        src, tgt, _ = synthetic.rigid_pair(N_SHARD, seed=53)
        reg = cpd.RigidCPD(src)
        reg._initialize(tgt)
        plan = reg._plan
        plan.set_moments_only(modes[rank])
        single, engine = [], []
        for _ in range(K_SHARD):
            plan.estep(W_SHARD)
            single.append(plan.last_estep_fused())
            engine.append(plan.last_estep_engine())
            reg._all_reduce_moments(plan)
            reg._device_mstep(plan)
        res = reg._result_from_params(plan.get_params())
        ret[rank] = dict(sigma2=float(res.sigma2), rot=np.array(res.transformation.rot), t=np.array(res.transformation.t),
                         single=single, engine=engine)
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("modes", [(1, 1), (1, 2)])
def test_single_sweeps_on_a_two_rank_shard(modes):
    """26 iterations with w = 0 on two ranks sharing the GPU (gloo): long enough for each rank to leave the matrix cores at
    its OWN iteration (fused sweep -> residual-form vector sweep; the decision goes by the rank's local pair counts).
    modes (1, 2): the ranks DISAGREE on the engine throughout - rank 0 runs single sweeps (MOMENTS[16] = tr Syy, [17..21] = 0),
    rank 1 two sweeps (all six entries of Syy) - and the all-reduced block still feeds the same rigid M-step (it reads the
    trace).  Result against the unsharded oracle."""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(2, _free_port(), ret, modes), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert a["single"] == [1] * K_SHARD
    assert b["single"] == ([1] * K_SHARD if modes[1] == 1 else [0] * K_SHARD)
    for r in (a, b):   # every rank hands over from the matrix cores once, for good, within the window
        assert r["engine"][0] == 1 and r["engine"][-1] == 0 and r["engine"] == sorted(r["engine"], reverse=True), r["engine"]
    assert a["sigma2"] == b["sigma2"] and np.array_equal(a["rot"], b["rot"])
    p, s2, _q = _shard_oracle()
    assert abs(a["sigma2"] - s2) <= TOL_SIGMA2 * s2, (a["sigma2"], s2)
    assert np.max(np.abs(a["rot"] - p["rot"])) <= TOL_TF
    assert np.max(np.abs(a["t"] - p["t"])) <= TOL_TF
    print("2-rank shard, modes %s: column engines rank 0 %s, rank 1 %s" % (modes, a["engine"], b["engine"]))
