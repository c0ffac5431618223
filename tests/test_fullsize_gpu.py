"""Parity at the BASELINE.json configurations THEMSELVES: the product (HIP path through the C ABI) and the CPU
oracle run on the exact inputs ``bench.py`` uses, for a handful of EM iterations, and must agree within the
north-star tolerances (transform 1e-4 relative, sigma2 1e-5 relative).

  C1  RigidCPD   N = M = 100 000   3 iterations from the identity (dense sweeps), 1 iteration continued from the GPU's
                                   own state in the MID regime (sigma2 < 3e-3: the matrix-core column pass with its
                                   per-wave tile masks next to the culled vector row pass) and 2 iterations continued
                                   from the state after 25 (late regime: 196 segments / 49 partial planes, ~97 % of
                                   the (wave, group) blocks culled) - oracle/cpd_estep_c.c, ~6 s per iteration
  C2  AffineCPD  N = M = 200 000   1 iteration from the identity, 1 in the mid regime, 1 from iteration 22 (~29 s each)
  C3  NonRigid   N = M = 12 000    3 iterations (largest M whose three M x M fp64 temporaries the numpy oracle holds
                                   comfortably), through the kernel factor (rank ~175) AND through the dense fallback
                                   (blocked Cholesky: 94 diagonal blocks / 24 outer panels)
  C3  NonRigid   N = M = 50 000    4 iterations, kernel factor against the dense fallback; and iteration 4 of BOTH solvers
                                   against the reference's own formulation solved by LAPACK on the host (dgesv on the
                                   20 GB float64 system built from the float32 G; self-skips below 48 GB of free RAM)
  C4  FilterReg  N = M = 500 000   5 iterations, 5 % outliers, sigma2 updated - oracle/filterreg_numpy.py on the C lattice

Reference lines: probreg/cpd.py:106-120 (driver), :71-88 (E-step), :160-192 / :219-244 / :284-303 (M-steps);
probreg/filterreg.py:120-147, :78-108, :158-196.
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_TF = 1e-4
TOL_SIGMA2 = 1e-5


def _oracle_iterations(kind, src, tgt, params, sigma2, k, w=0.0):
    """k EM iterations of the reference algorithm from an explicit state (C E-step + numpy M-step, all fp64)."""
    from oracle import cpd_c, cpd_numpy as co

    for _ in range(k):
        ts = co.transform(kind, params, src)
        es = co.EstepResult(*cpd_c.expectation_step(ts, tgt, sigma2, w))
        if kind == "rigid":
            params, sigma2, q = co.mstep_rigid(src, tgt, es)
        else:
            params, sigma2, q = co.mstep_affine(src, tgt, es)
    return params, sigma2, q


def _check(kind, res, params, sigma2, q):
    tr = res.transformation
    if kind == "rigid":
        assert rel_err(tr.rot, params["rot"]) < TOL_TF
        assert abs(tr.scale - params["scale"]) < TOL_TF * abs(params["scale"])
    else:
        assert rel_err(tr.b, params["b"]) < TOL_TF
    assert np.max(np.abs(tr.t - params["t"])) < TOL_TF * max(1.0, np.max(np.abs(params["t"])))
    assert abs(res.sigma2 - sigma2) <= TOL_SIGMA2 * abs(sigma2)
    assert abs(res.q - q) <= 1e-4 * abs(q)


def _state_as_oracle_params(kind, res):
    tr = res.transformation
    if kind == "rigid":
        return dict(rot=tr.rot.copy(), t=tr.t.copy(), scale=float(tr.scale))
    return dict(b=tr.b.copy(), t=tr.t.copy())


@pytest.mark.parametrize("config", ["C1_rigid_100k", "C2_affine_200k"])
def test_cpd_bench_config_vs_oracle_dense_and_late(config):
    from oracle import cpd_numpy as co
    from probreg_amd import cpd, synthetic

    if config.startswith("C1"):
        kind, n, k_dense, k_warm, k_late = "rigid", 100000, 3, 25, 2
        src, tgt, _ = synthetic.rigid_pair(n, seed=0)
        reg = cpd.RigidCPD(src)
        ident = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0)
    else:
        kind, n, k_dense, k_warm, k_late = "affine", 200000, 1, 22, 1
        src, tgt, _ = synthetic.affine_pair(n, seed=0)
        reg = cpd.AffineCPD(src)
        ident = dict(b=np.identity(3), t=np.zeros(3))

    # ---- from the identity: every pair is evaluated (dense regime) ----
    res = reg.registration(tgt, w=0.0, maxiter=k_dense, tol=-1.0)
    # what is held to the oracle below, by name: C1 - the FUSED single sweep on the matrix cores (no row pass: nothing reported
    # for one); C2 - the matrix-core column pass and the LEAN matrix-core row pass (no residual sums)
    if kind == "rigid":
        assert reg._plan.last_estep_fused() == 1 and reg._plan.last_estep_engines() == (1, 0) and reg._plan.last_estep_lean() == 0
        reg._plan.set_moments_only(1)   # the registration's own E-steps below as well: single sweeps in every regime
    else:
        assert reg._plan.last_estep_fused() == 0 and reg._plan.last_estep_engines() == (1, 1) and reg._plan.last_estep_lean() == 1
    s2_0 = co.squared_kernel_sum_closed_form(src, tgt)
    p, s2, q = _oracle_iterations(kind, src, tgt, ident, s2_0, k_dense)
    _check(kind, res, p, s2, q)

    # ---- mid regime: the matrix-core column pass with chunk / tile masks hands over to the culled vector row pass ----
    plan = reg._plan
    done = k_dense
    while True:
        mid = reg._result_from_params(plan.get_params())
        # (rigid: the fused sweep hands over to the vector pipe earlier than the matrix-core column pass of the two-sweep engine
        # does - C1: EM iteration 12, sigma2 = 2.7e-3 - so its masked matrix-core iteration is taken at sigma2 < 8e-3, iteration 10)
        if (mid.sigma2 < (8e-3 if kind == "rigid" else 3e-3) and done >= 8) or done >= k_warm - 2:
            break
        plan.estep(0.0)
        reg._device_mstep(plan)
        done += 1
    assert 1e-4 < mid.sigma2 < 8e-3 and done < k_warm - 2, (done, mid.sigma2)
    plan.estep(0.0)
    assert plan.last_estep_engine() == 1  # this iteration's column pass ran on the matrix cores (tile-mask regime)
    col_pairs, _row_pairs = plan.pair_counts()
    assert col_pairs < 0.9 * float(n) * n  # ... and its chunk / tile masks did skip blocks
    reg._device_mstep(plan)
    done += 1
    res = reg._result_from_params(plan.get_params())
    p, s2, q = _oracle_iterations(kind, src, tgt, _state_as_oracle_params(kind, mid), mid.sigma2, 1)
    _check(kind, res, p, s2, q)

    # ---- late regime: continue on the GPU from its own iteration-k_warm state, hand that state to the oracle ----
    for _ in range(k_warm - done):
        plan.estep(0.0)
        reg._device_mstep(plan)
    warm = reg._result_from_params(plan.get_params())
    assert warm.sigma2 < 1e-3  # this IS the culled regime (sigma2_0 is ~0.3)
    for _ in range(k_late):
        plan.estep(0.0)
        reg._device_mstep(plan)
    if kind == "rigid":   # the late regime of C1 is the residual-form single sweep over the work queue
        assert plan.last_estep_fused() == 1 and plan.last_estep_engines() == (0, 0)
    res = reg._result_from_params(plan.get_params())
    p, s2, q = _oracle_iterations(kind, src, tgt, _state_as_oracle_params(kind, warm), warm.sigma2, k_late)
    _check(kind, res, p, s2, q)


@pytest.fixture(scope="module")
def c3_style_oracle():
    from oracle import cpd_numpy as co
    from probreg_amd import synthetic

    src, tgt = synthetic.nonrigid_pair(12000, seed=0)
    p, s2, q, _ = co.registration("nonrigid", src, tgt, maxiter=3, tol=-1.0, closed_form_init=True)
    return src, tgt, s2, co.transform("nonrigid", p, src, co.rbf_kernel(src, src, 2.0))


@pytest.mark.parametrize("solver", ["factor", "dense"])
def test_nonrigid_c3_style_12k_vs_oracle(c3_style_oracle, solver):
    """C3's kernels at the largest size the numpy oracle's LAPACK solve holds, beta = lmd = 2 as in the config.  "factor":
    the default path (G = F F^T, r x r reduced system); "dense": the fallback - M = N = 12 000 is 94 diagonal blocks of the
    blocked fp64 Cholesky and 24 outer 512-column panels (C3 itself: 391 / 98)."""
    from probreg_amd import cpd

    src, tgt, s2, want = c3_style_oracle

    class Dense(cpd.NonRigidCPD):
        _solver_mode = 0

    reg = (cpd.NonRigidCPD if solver == "factor" else Dense)(src)
    res = reg.registration(tgt, maxiter=3, tol=-1.0)
    assert (reg._plan.nonrigid_rank() > 0) == (solver == "factor")
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2
    got = res.transformation.transform(src)
    assert np.max(np.abs(got - want)) < TOL_TF * np.max(np.abs(want - want.mean(0)))


def test_nonrigid_c3_full_size_factor_vs_dense_fallback():
    """C3 itself (N = M = 50 000) is beyond the numpy oracle (three 20 GB temporaries per M-step).  The chain of evidence at
    this size: both solvers match the oracle at 12k (above); here they must match EACH OTHER at 50k over 4 iterations - the
    dense fallback being the path that evaluates the reference's own float32 M x M matrix (391 Cholesky blocks)."""
    from probreg_amd import cpd, synthetic

    src, tgt = synthetic.nonrigid_pair(50000, seed=0)

    class Dense(cpd.NonRigidCPD):
        _solver_mode = 0

    out = []
    for cls in (cpd.NonRigidCPD, Dense):
        reg = cls(src)
        res = reg.registration(tgt, maxiter=4, tol=-1.0)
        out.append((res.sigma2, res.transformation.transform(src), reg._plan.nonrigid_rank()))
        del reg, res
    assert out[0][2] > 0 and out[1][2] == 0
    ext = np.max(np.abs(out[1][1] - out[1][1].mean(0)))
    assert np.max(np.abs(out[0][1] - out[1][1])) < TOL_TF * ext
    assert abs(out[0][0] - out[1][0]) <= TOL_SIGMA2 * out[1][0]


def test_nonrigid_c3_full_size_vs_lapack():
    """C3 at its own size against the reference's formulation (cpd.py:284-303) solved on the HOST: EM iteration 4, continued
    from the GPU's state after 3, through oracle/cpd_estep_c.c (E-step) and LAPACK dgesv on the M x M float64 system built
    from the float32 kernel matrix - 20 GB, ~1 min on the GPU box's cores.  Both GPU solvers (kernel factor, dense
    Cholesky) run the same iteration from the same state."""
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 48 * 2 ** 30:
        pytest.skip("needs ~30 GB of free host memory for the 50 000 x 50 000 float64 system")
    from scipy.linalg import lapack

    from oracle import cpd_c
    from probreg_amd import cpd, synthetic

    m, beta, lmd = 50000, 2.0, 2.0
    src, tgt = synthetic.nonrigid_pair(m, seed=0)
    reg = cpd.NonRigidCPD(src, beta=beta, lmd=lmd)
    assert not np.any(reg._origin)  # the plan holds the float32 cast of the caller's coordinates, like the reference's G
    res3 = reg.registration(tgt, maxiter=3, tol=-1.0)
    plan = reg._plan
    assert plan.nonrigid_rank() > 0
    state3, w3, s2_3 = plan.get_params(), plan.get_w(), res3.sigma2

    # ---- the oracle's iteration 4 from that state ----
    ts = src + cpd_c.nonrigid_gw(src, beta, w3)                      # transformation.py:101-102
    pt1, p1, px, n_p = cpd_c.expectation_step(ts, tgt, s2_3, 0.0)    # cpd.py:71-88
    a = cpd_c.nonrigid_lhs(src, beta, p1, lmd * s2_3)                # (p1 * g).T + lmd sigma2 I, cpd.py:297
    rhs = np.asfortranarray(px - src * p1[:, None])
    _lu, _piv, w4, info = lapack.dgesv(a, rhs, overwrite_a=1, overwrite_b=1)
    assert info == 0
    del a, _lu
    t4 = src + cpd_c.nonrigid_gw(src, beta, w4)
    tr_xp1x = float(np.sum(pt1 * np.sum(tgt * tgt, axis=1)))
    tr_pxt = float(np.sum(px * t4))
    tr_tpt = float(np.sum(p1 * np.sum(t4 * t4, axis=1)))
    s2_4 = (tr_xp1x - 2.0 * tr_pxt + tr_tpt) / (n_p * 3)             # cpd.py:299-302
    ext = np.max(np.abs(t4 - t4.mean(0)))

    # ---- the same iteration on the GPU: kernel factor (the plan is at state 3), then the dense fallback ----
    def gpu_iteration(pl):
        pl.estep(0.0)
        pl.mstep_nonrigid(lmd)
        return float(pl.get_params()[13]), pl.nonrigid_apply()

    got = {"factor": gpu_iteration(plan)}
    del reg, plan

    class Dense(cpd.NonRigidCPD):
        _solver_mode = 0

    regd = Dense(src, beta=beta, lmd=lmd)
    regd._initialize(tgt)
    assert regd._plan.nonrigid_rank() == 0
    regd._plan.set_params(state3)
    regd._plan.set_w(w3)
    got["dense"] = gpu_iteration(regd._plan)
    for solver, (s2, t_gpu) in got.items():
        assert abs(s2 - s2_4) <= TOL_SIGMA2 * s2_4, solver
        assert np.max(np.abs(t_gpu - t4)) < TOL_TF * ext, solver


def test_filterreg_c4_500k_vs_oracle():
    """C4 at full size against the CPU lattice (bit-identical to the reference's vendored permutohedral.cpp)."""
    from oracle import cpd_numpy as co, filterreg_numpy as fo
    from probreg_amd import filterreg, synthetic

    src, tgt, _ = synthetic.filterreg_pair(500000, seed=0)
    # the reference's initialiser builds the dense M x N float32 matrix (2.5e11 entries); both sides get its
    # closed form rounded to float32 (what mu.squared_kernel_sum returns) as an explicit sigma2
    s2_0 = float(np.float32(co.squared_kernel_sum_closed_form(src, tgt)))
    k = 5
    rot, t, s2, q, n_iter = fo.registration(src, tgt, sigma2=s2_0, update_sigma2=True, w=0.05, maxiter=k, tol=-1.0)
    assert n_iter == k
    res = filterreg.registration_filterreg(src, tgt, sigma2=s2_0, update_sigma2=True, w=0.05, maxiter=k, tol=-1.0)
    assert rel_err(res.transformation.rot, rot) < TOL_TF
    assert np.max(np.abs(res.transformation.t - t)) < TOL_TF
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2
    assert abs(res.q - q) <= 1e-4 * abs(q)
