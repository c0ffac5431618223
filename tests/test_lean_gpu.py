"""The LEAN matrix-core row pass (k_rowpass_mfma<LEAN> leaves out the residual sums sum_n P |x_n - o|^2 and
the M-step takes sum_n pt1_n |x_n|^2 from the column side, k_colfinal's partials -> k_xpx_columns) by name: it is ON by
default for the first EM iterations of C1 / C2 (tests/test_fullsize_gpu.py asserts that on the compared iteration); here it
is FORCED (prg_cpd_set_lean_factor(1e30): lean wherever the matrix-core row pass runs) far beyond the amplification
mean |x|^2 / (sigma2 D) <= 64 where the default allows it, and sigma2 after the M-step (cpd.py:186-191: sigma2 and q from
the moments) is held to 1e-5 of the fp64 oracle's - with w = 0 and w = 0.1, and on a 2-rank shard."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_SIGMA2 = 1e-5
TOL_TF = 1e-4


def _oracle_step(kind, src, tgt, st, w):
    from oracle import cpd_c, cpd_numpy as co

    tr = st.transformation
    p = dict(rot=tr.rot, t=tr.t, scale=float(tr.scale)) if kind == "rigid" else dict(b=tr.b, t=tr.t)
    es = co.EstepResult(*cpd_c.expectation_step(co.transform(kind, p, src), tgt, st.sigma2, w))
    return (co.mstep_rigid if kind == "rigid" else co.mstep_affine)(src, tgt, es)


@pytest.mark.parametrize("w", [0.0, 0.1])
def test_forced_lean_pass_along_a_100k_rigid_registration(w):
    """C1's clouds (w = 0; half of C1's points with w = 0.1 - the oracle's cost goes with n^2); both sweeps pinned to the matrix
    cores, the row pass lean in EVERY iteration; at the iterations listed the GPU's state before the iteration goes to the C
    oracle and the two M-step results are compared."""
    from probreg_amd import cpd, synthetic

    n = 100000 if w == 0.0 else 50000
    src, tgt, _ = synthetic.rigid_pair(n, seed=0)
    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_dense_engine(2)
    plan.set_lean_factor(1e30)
    mean_x2 = float(np.mean(np.sum((tgt - tgt.mean(0)) ** 2, axis=1)))
    checked, worst, top_amp = 0, 0.0, 0.0
    for it in range(14):
        st = reg._result_from_params(plan.get_params())
        amp = mean_x2 / (3.0 * st.sigma2)
        plan.estep(w)
        assert plan.last_estep_lean() == 1, it
        assert plan.last_estep_engines() == (1, 1)
        reg._device_mstep(plan)
        if it in ((0, 12, 13) if w == 0.0 else (0, 4, 8, 10, 12, 13)):
            out = reg._result_from_params(plan.get_params())
            p, s2, q = _oracle_step("rigid", src, tgt, st, w)
            err = abs(out.sigma2 - s2) / s2
            assert err <= TOL_SIGMA2, (it, amp, err)
            assert np.max(np.abs(out.transformation.rot - p["rot"])) <= TOL_TF
            assert abs(out.q - q) <= 1e-4 * abs(q)
            worst, top_amp, checked = max(worst, err), max(top_amp, amp), checked + 1
    assert checked == (3 if w == 0.0 else 6) and top_amp >= 64.0, (checked, top_amp)  # held up to an amplification of >= 64 (default limit: 16)
    print("forced lean pass, w = %g: worst sigma2 error %.2e up to amplification %.0f" % (w, worst, top_amp))


def test_default_lean_window_and_switch_off():
    """Default factor (64): the matrix-core row passes of C1 run lean up to that amplification, and the flag is off once the row
    pass has gone to the vector pipe; a smaller factor (16, round 3's default) can only turn it off earlier; factor 0 never runs it; all give the
    oracle's sigma2 (checked on the last iteration whose row pass runs on the matrix cores under every setting)."""
    from probreg_amd import cpd, synthetic

    src, tgt, _ = synthetic.rigid_pair(100000, seed=0)
    seen = {}
    for factor in (-1.0, 16.0, 0.0):
        reg = cpd.RigidCPD(src)
        reg._initialize(tgt)
        plan = reg._plan
        plan.set_lean_factor(factor)
        flags, rows = [], []
        for it in range(15):
            st = reg._result_from_params(plan.get_params()) if it == 6 and factor != 16.0 else None
            plan.estep(0.0)
            flags.append(plan.last_estep_lean())
            rows.append(plan.last_estep_engines()[1])
            reg._device_mstep(plan)
            if st is not None:
                _, s2, _ = _oracle_step("rigid", src, tgt, st, 0.0)
                out = reg._result_from_params(plan.get_params())
                assert abs(out.sigma2 - s2) <= TOL_SIGMA2 * s2
        seen[factor] = (flags, rows)
    flags, rows = seen[-1.0]
    assert flags[:7] == [1] * 7 and flags[-1] == 0 and rows[-1] == 0, (flags, rows)
    assert flags == sorted(flags, reverse=True)                      # ... never back on
    # lean only where the matrix-core row pass ran (the row pass may stay on the matrix cores past amplification 64: not lean there)
    assert all(f <= r for f, r in zip(flags, rows)), (flags, rows)
    f16, r16 = seen[16.0]
    assert f16[:6] == [1] * 6 and all(a <= b for a, b in zip(f16, flags)) and all(f <= r for f, r in zip(f16, r16)), (f16, r16)
    assert seen[0.0][0] == [0] * 15 and seen[0.0][1][:7] == [1] * 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_SHARD, K_SHARD, W_SHARD = 40000, 9, 0.1


def _shard_worker(rank, world, port, ret):
    import torch
    import torch.distributed as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from probreg_amd import cpd, synthetic

        src, tgt, _ = synthetic.rigid_pair(N_SHARD, seed=11)
        reg = cpd.RigidCPD(src)
        reg._initialize(tgt)
        plan = reg._plan
        assert plan.n < N_SHARD  # a shard
        plan.set_dense_engine(2)
        plan.set_lean_factor(1e30)
        lean = []
        for _ in range(K_SHARD):
            plan.estep(W_SHARD)
            lean.append(plan.last_estep_lean())
            reg._all_reduce_moments(plan)   # k_xpx_columns has scaled by the LOCAL row / column sums before this
            reg._device_mstep(plan)
        res = reg._result_from_params(plan.get_params())
        ret[rank] = dict(sigma2=float(res.sigma2), q=float(res.q), rot=np.array(res.transformation.rot), lean=lean)
    finally:
        tdist.destroy_process_group()


def test_forced_lean_pass_on_a_two_rank_shard():
    """Lean + w > 0 + a 2-rank shard: every rank scales its column-side sum by its LOCAL (row sums / column sums) - both over
    the same pairs, all m x the rank's n - before the all-reduce.  Two processes on this GPU (gloo), the real kernels; the
    result against the unsharded fp64 oracle."""
    import torch.multiprocessing as mp

    from oracle import cpd_numpy as co
    from probreg_amd import synthetic

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert a["lean"] == [1] * K_SHARD and b["lean"] == [1] * K_SHARD
    assert a["sigma2"] == b["sigma2"] and np.array_equal(a["rot"], b["rot"])
    src, tgt, _ = synthetic.rigid_pair(N_SHARD, seed=11)
    from oracle import cpd_c

    p, s2 = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0), co.squared_kernel_sum_closed_form(src, tgt)
    for _ in range(K_SHARD):  # the reference's loop (cpd.py:110-113) with the C / OpenMP E-step
        es = co.EstepResult(*cpd_c.expectation_step(co.transform("rigid", p, src), tgt, s2, W_SHARD))
        p, s2, _q = co.mstep_rigid(src, tgt, es)
    assert abs(a["sigma2"] - s2) <= TOL_SIGMA2 * s2, (a["sigma2"], s2)
    assert np.max(np.abs(a["rot"] - p["rot"])) <= TOL_TF
