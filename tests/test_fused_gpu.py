"""The FUSED single sweep of a rigid EM iteration (csrc/cpd_sweeps_mfma.hip k_colpass_mfma<FUSED>,
csrc/cpd.hip k_colfinal_fused / k_fused_final): while sigma2 is large the rigid M-step's 23 moments (cpd.py:160-192) are
taken from per-column sums of ONE sweep over the pairs instead of a column pass and a row pass.  Held to the two-sweep
engine from the same state, to the fp64 oracle along registrations (north-star tolerances), and to the error behaviour of
the interface (no per-point p1 / px after such an E-step)."""
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


@pytest.mark.parametrize("n,m,w", [(40000, 40000, 0.0), (30011, 45007, 0.1), (9000, 12000, 0.0)])
def test_fused_sweep_moments_equal_the_two_sweep_engine(n, m, w):
    from probreg_amd import _lib, cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=31)
    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_moments_only(2)
    for _ in range(2):
        plan.estep(w)
        plan.mstep(_lib.PRG_TF_RIGID, True)
    state = plan.get_params()
    out = {}
    for mode in (2, 1):
        plan.set_moments_only(mode)
        plan.set_dense_engine(1)   # (resets the switch's memory)
        plan.set_params(state)
        plan.estep(w)
        plan.set_params(state)
        plan.estep(w)
        assert plan.last_estep_fused() == (1 if mode == 1 else 0)
        assert plan.last_estep_engines()[0] == 1
        mom = plan.get_moments()
        pt1 = plan.get_estep_pt1()
        plan.mstep(_lib.PRG_TF_RIGID, True)
        out[mode] = (mom, pt1, plan.get_params())
    a, b = out[2][0], out[1][0]
    n_p = a[0]
    assert abs(a[0] - b[0]) < 2e-6 * n_p
    assert np.max(np.abs(a[1:16] - b[1:16])) < 2e-6 * n_p           # Sx, Sy, Sxy
    assert abs((a[16] + a[19] + a[21]) - b[16]) < 2e-6 * n_p and np.all(b[17:22] == 0.0)   # tr Syy
    assert abs(a[22] - b[22]) < 2e-6 * n_p                          # sum pt1 |x|^2
    assert np.max(np.abs(out[2][1] - out[1][1])) < 1e-6             # pt1
    # ... and the M-step that follows gives the same transformation and sigma2
    pa, pb = out[2][2], out[1][2]
    assert np.max(np.abs(pa[:13] - pb[:13])) < 2e-6
    assert abs(pa[13] - pb[13]) <= 5e-6 * pa[13]
    # no per-point arrays after a fused E-step: the interface says so
    plan.set_params(state)
    plan.estep(w)
    with pytest.raises(_lib.ProbregHipError, match="fused single sweep"):
        plan.get_estep()


@pytest.mark.parametrize("case", ["c1_50k", "scale_fixed_w", "two_d", "init_rot"])
def test_registration_through_the_fused_sweep_matches_the_oracle(case):
    """`registration` with tol < 0 runs prg_cpd_iterate: rigid iterations take the fused sweep while they may.  Against the
    reference's loop (C E-step + numpy M-step, fp64)."""
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    kw, init = dict(), None
    if case == "c1_50k":   # (C1 itself, 100k from the identity: tests/test_fullsize_gpu.py)
        src, tgt, _ = synthetic.rigid_pair(50000, seed=0)
        k, w = 4, 0.0
    elif case == "scale_fixed_w":
        src, tgt, _ = synthetic.rigid_pair(30000, m=26000, seed=7)
        k, w, kw = 6, 0.2, dict(update_scale=False)
    elif case == "two_d":
        src, tgt, _ = synthetic.rigid_pair(20000, m=24000, seed=9)
        src, tgt = src[:, :2].copy(), tgt[:, :2].copy()
        k, w = 5, 0.0
    else:
        src, tgt, _ = synthetic.rigid_pair(25000, seed=11)
        rot0 = synthetic.rot_zx(20.0, -8.0)
        init = dict(rot=rot0, t=np.array([0.05, 0.0, -0.02]), scale=1.1)
        k, w = 3, 0.05
    reg = cpd.RigidCPD(src, tf_init_params=init or {}, **kw)
    res = reg.registration(tgt, w=w, maxiter=k, tol=-1.0)
    assert reg._plan.last_estep_fused() == 1
    dim = src.shape[1]
    p0 = dict(rot=np.identity(dim), t=np.zeros(dim), scale=1.0) if init is None else dict(init)
    s2_0 = co.squared_kernel_sum_closed_form(src, tgt)
    p, s2, q = _oracle_iterations(src, tgt, p0, s2_0, k, w, kw.get("update_scale", True))
    tr = res.transformation
    assert np.max(np.abs(tr.rot - p["rot"])) < TOL_TF
    assert np.max(np.abs(tr.t - p["t"])) < TOL_TF * max(1.0, np.max(np.abs(p["t"])))
    assert abs(tr.scale - p["scale"]) < TOL_TF * p["scale"]
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2
    assert abs(res.q - q) <= 1e-4 * abs(q)


def test_a_non_orthonormal_starting_matrix_keeps_the_two_sweeps():
    """The column-side sums are mapped back through s R: a starting `rot` that is not a rotation (the reference takes any
    matrix) must not take the fused path in the first iteration."""
    from probreg_amd import _lib, cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(20000, seed=13)
    bad = np.identity(3)
    bad[0, 1] = 0.05
    reg = cpd.RigidCPD(src, tf_init_params=dict(rot=bad))
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_moments_only(1)
    plan.estep(0.0)
    assert plan.last_estep_fused() == 0
    plan.mstep(_lib.PRG_TF_RIGID, True)


@pytest.mark.parametrize("forced", [False, True])
def test_fused_sweep_along_a_100k_registration(forced):
    """C1's clouds, every E-step a single sweep: at the iterations listed the GPU's state before the iteration goes to the C
    oracle and the two M-step results are compared.  Default: 50 iterations - the fused matrix-core sweep while the matrix-core
    column pass would run, then, once and for good, the residual-form sweep of the vector pipe down to the noise
    floor (amplification > 1e3: the residual form has no limit).  Forced (both engines pinned to the matrix cores,
    prg_cpd_set_fused_factor(1e30)): the fused sweep's sigma2 held to 1e-5 beyond its default factor of 256."""
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(100000, seed=0)
    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_moments_only(1)
    if forced:
        plan.set_dense_engine(2)
        plan.set_fused_factor(1e30)
    mean_x2 = float(np.mean(np.sum((tgt - tgt.mean(0)) ** 2, axis=1)))
    single, engines, worst, top_amp, worst_v, top_amp_v = [], [], 0.0, 0.0, 0.0, 0.0
    for it in range(18 if forced else 50):
        st = reg._result_from_params(plan.get_params())
        amp = mean_x2 / (3.0 * st.sigma2)
        plan.estep(0.0)
        single.append(plan.last_estep_fused())
        engines.append(plan.last_estep_engine())
        assert plan.last_estep_engines()[1] == 0 and plan.last_estep_lean() == 0   # no row pass ran: nothing reported for it
        reg._device_mstep(plan)
        if it in ((0, 16, 17) if forced else (0, 11, 14, 30, 49)):
            out = reg._result_from_params(plan.get_params())
            tr = st.transformation
            es = co.EstepResult(*cpd_c.expectation_step(co.transform("rigid", dict(rot=tr.rot, t=tr.t, scale=float(tr.scale)), src),
                                                        tgt, st.sigma2, 0.0))
            p, s2, q = co.mstep_rigid(src, tgt, es)
            err = abs(out.sigma2 - s2) / s2
            assert err <= TOL_SIGMA2, (it, amp, engines[-1], err)
            assert np.max(np.abs(out.transformation.rot - p["rot"])) <= TOL_TF
            assert np.max(np.abs(out.transformation.t - p["t"])) <= TOL_TF
            if engines[-1]:
                worst, top_amp = max(worst, err), max(top_amp, amp)
            else:
                worst_v, top_amp_v = max(worst_v, err), max(top_amp_v, amp)
    assert single == [1] * len(single), single
    if forced:
        assert engines == [1] * 18 and top_amp >= 500.0, (engines, top_amp)
    else:
        assert engines[:12] == [1] * 12 and engines[-1] == 0, engines
        assert engines == sorted(engines, reverse=True)   # hands over once, for good
        assert top_amp >= 20.0 and top_amp_v >= 1000.0, (top_amp, top_amp_v)
    print("single sweeps (forced %d): fused worst sigma2 error %.2e up to amplification %.0f; residual form %.2e up to %.0f; column engines %s"
          % (forced, worst, top_amp, worst_v, top_amp_v, engines))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_SHARD, K_SHARD, W_SHARD = 40000, 7, 0.1


def _shard_worker(rank, world, port, ret):
    import torch
    import torch.distributed as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from probreg_amd import cpd, synthetic

        src, tgt, _ = synthetic.rigid_pair(N_SHARD, seed=15)
        reg = cpd.RigidCPD(src)
        reg._initialize(tgt)
        plan = reg._plan
        plan.set_moments_only(1)
        fused = []
        for _ in range(K_SHARD):
            plan.estep(W_SHARD)
            fused.append(plan.last_estep_fused())
            reg._all_reduce_moments(plan)   # partial moments of this rank's columns, mapped back before the all-reduce (linear)
            reg._device_mstep(plan)
        res = reg._result_from_params(plan.get_params())
        ret[rank] = dict(sigma2=float(res.sigma2), rot=np.array(res.transformation.rot), fused=fused)
    finally:
        tdist.destroy_process_group()


def test_fused_sweep_on_a_two_rank_shard():
    import torch.multiprocessing as mp

    from oracle import cpd_numpy as co
    from probreg_amd import synthetic

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert a["fused"] == [1] * K_SHARD and b["fused"] == [1] * K_SHARD
    assert a["sigma2"] == b["sigma2"] and np.array_equal(a["rot"], b["rot"])
    src, tgt, _ = synthetic.rigid_pair(N_SHARD, seed=15)
    p, s2, _q = _oracle_iterations(src, tgt, dict(rot=np.identity(3), t=np.zeros(3), scale=1.0),
                                   co.squared_kernel_sum_closed_form(src, tgt), K_SHARD, W_SHARD)
    assert abs(a["sigma2"] - s2) <= TOL_SIGMA2 * s2, (a["sigma2"], s2)
    assert np.max(np.abs(a["rot"] - p["rot"])) <= TOL_TF
