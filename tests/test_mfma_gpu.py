"""The matrix-core (f32 MFMA) E-step sweeps of the dense regime (csrc/cpd_sweeps_mfma.hip) against the
oracle and against the vector-pipe sweeps on the same state.  Reference: probreg/cpd.py:71-88."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_TF = 1e-4
TOL_SIGMA2 = 1e-5


def _plan_with_state(src, tgt, k_warm, engine):
    from probreg_amd import _lib, cpd

    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_dense_engine(engine)
    for _ in range(k_warm):
        plan.estep(0.0)
        plan.mstep(_lib.PRG_TF_RIGID, True)
    return reg, plan


@pytest.mark.parametrize("n,m,k_warm,w", [(6000, 6000, 1, 0.0), (9000, 5000, 3, 0.1), (3001, 4097, 2, 0.0)])
def test_mfma_estep_matches_oracle_and_vector_sweeps(n, m, k_warm, w):
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import synthetic

    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=17)
    reg, plan = _plan_with_state(src, tgt, k_warm, 0)  # warm-up on the vector-pipe sweeps: both engines see ONE state
    state = plan.get_params()
    out = {}
    for engine in (0, 2):
        plan.set_dense_engine(engine)
        plan.set_params(state)
        plan.estep(w)
        assert plan.last_estep_engine() == (1 if engine == 2 else 0)
        out[engine] = (plan.get_moments(), plan.get_estep())
    res = reg._result_from_params(state)
    ts = co.transform("rigid", dict(rot=res.transformation.rot, t=res.transformation.t, scale=res.transformation.scale), src)
    pt1, p1, px, n_p = cpd_c.expectation_step(ts, tgt, res.sigma2, w)
    for engine in (0, 2):
        mom, (g_pt1, g_p1, g_px) = out[engine]
        g_px = g_px + np.outer(g_p1, reg._cx)  # the plan works on the centred target
        assert abs(mom[0] - n_p) < 2e-6 * n_p, engine
        assert np.max(np.abs(g_pt1 - pt1)) < 2e-5, engine
        assert np.max(np.abs(g_p1 - p1)) < 2e-5 * max(1.0, p1.max()), engine
        assert np.max(np.abs(g_px - px)) < 2e-5 * max(1.0, np.abs(px).max()), engine
    # the 23 moments the M-step consumes are sums over ~n_p points of O(1) terms (some cancel to ~0 in the centred
    # frame): both engines within 2e-6 n_p of each other
    a, b = out[0][0][:23], out[2][0][:23]
    assert np.max(np.abs(a - b)) < 2e-6 * n_p


def test_auto_engine_registration_matches_oracle_and_switches_engines():
    """Engine mode 1: matrix-core column pass while |kk| * extent^2 is below the bound, culled vector-pipe sweeps
    afterwards; the whole trajectory stays within the north-star tolerances of the fp64 oracle."""
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import _lib, cpd, synthetic

    n, k = 20000, 22
    src, tgt, _ = synthetic.rigid_pair(n, seed=4)
    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_dense_engine(1)
    engines = []
    for _ in range(k):
        plan.estep(0.0)
        engines.append(plan.last_estep_engine())
        plan.mstep(_lib.PRG_TF_RIGID, True)
    res = reg._result_from_params(plan.get_params())
    assert engines[0] == 1            # first E-step: no column minima yet, but sigma2 is large enough for zero offsets
    assert sum(engines) >= 4          # the dense regime's column passes ran on the matrix cores ...
    assert engines[-1] == 0           # ... and the late regime did not
    first_off = engines.index(0)
    assert all(e == 0 for e in engines[first_off:])  # once left, never re-entered
    params = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0)
    sigma2 = co.squared_kernel_sum_closed_form(src, tgt)
    for _ in range(k):
        es = co.EstepResult(*cpd_c.expectation_step(co.transform("rigid", params, src), tgt, sigma2, 0.0))
        params, sigma2, q = co.mstep_rigid(src, tgt, es)
    assert rel_err(res.transformation.rot, params["rot"]) < TOL_TF
    assert np.max(np.abs(res.transformation.t - params["t"])) < TOL_TF
    assert abs(res.transformation.scale - params["scale"]) < TOL_TF * params["scale"]
    assert abs(res.sigma2 - sigma2) <= TOL_SIGMA2 * sigma2


def test_forced_matrix_core_engine_through_a_whole_dense_phase_2d():
    """2-D clouds (z = 0 plane) and an affine M-step on the matrix-core sweeps, forced for 8 iterations."""
    from oracle import cpd_numpy as co
    from probreg_amd import _lib, cpd

    rng = np.random.default_rng(3)
    th = np.linspace(0.0, 2.0 * np.pi, 3000, endpoint=False)
    src = np.stack([np.cos(th) * (1.0 + 0.3 * np.cos(3 * th)), 0.6 * np.sin(th)], axis=1) + rng.normal(0, 0.01, (3000, 2))
    a = np.array([[1.05, 0.1], [-0.08, 0.93]])
    tgt = src[rng.permutation(3000)[:2500]] @ a.T + np.array([0.05, -0.03]) + rng.normal(0, 0.01, (2500, 2))
    src = src.astype(np.float32).astype(np.float64)
    tgt = tgt.astype(np.float32).astype(np.float64)
    reg = cpd.AffineCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_dense_engine(2)
    used = 0
    for _ in range(8):
        plan.estep(0.05)
        used += plan.last_estep_engine()
        plan.mstep(_lib.PRG_TF_AFFINE, True)
    assert used == 8  # including the first E-step (zero offsets: sigma2 is large)
    res = reg._result_from_params(plan.get_params())
    p, s2, q, _ = co.registration("affine", src, tgt, w=0.05, maxiter=8, tol=-1.0, closed_form_init=True)
    assert rel_err(res.transformation.b, p["b"]) < TOL_TF
    assert np.max(np.abs(res.transformation.t - p["t"])) < TOL_TF
    assert abs(res.sigma2 - s2) <= TOL_SIGMA2 * s2


def test_engine_switch_goes_by_pair_counts_on_a_non_surface_cloud():
    """The switch out of the matrix-core sweeps is taken from what those sweeps evaluated one E-step back, not
    from sigma2: on a 10:1:1 box filled uniformly (no surface anywhere) it still starts on the matrix cores, the row pass
    leaves no later than the column pass, nothing is re-entered, the evaluated pairs fall monotonically while the matrix
    cores run, and the registration matches the fp64 oracle's trajectory."""
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import _lib, cpd, synthetic

    n, k = 32768, 18
    rng = np.random.default_rng(11)
    src = rng.random((n, 3)) * np.array([10.0, 1.0, 1.0])
    tgt = (src @ synthetic.rot_zx(12.0, 5.0).T + np.array([0.05, -0.03, 0.02]) + 0.004 * rng.standard_normal((n, 3)))[rng.permutation(n)]
    src = src.astype(np.float32).astype(np.float64)
    tgt = tgt.astype(np.float32).astype(np.float64)
    reg = cpd.RigidCPD(src)
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_dense_engine(1)
    engines, pairs = [], []
    for _ in range(k):
        plan.estep(0.0)
        engines.append(plan.last_estep_engines())
        pairs.append(plan.pair_counts())
        plan.mstep(_lib.PRG_TF_RIGID, True)
    res = reg._result_from_params(plan.get_params())
    col = [e[0] for e in engines]
    row = [e[1] for e in engines]
    assert engines[0] == (1, 1)
    assert engines[-1] == (0, 0)                      # sigma2 ~ 0.03 by now: a ball of 13 sigma holds a fraction of the box
    assert 0 in col and 0 in row
    col_off, row_off = col.index(0), row.index(0)
    assert 2 <= row_off <= col_off                    # dense for a while; the row pass leaves first
    assert all(e == 0 for e in col[col_off:]) and all(e == 0 for e in row[row_off:])
    on = [p[0] for p, e in zip(pairs, col) if e == 1]
    assert all(b <= a * 1.02 for a, b in zip(on, on[1:]))     # fewer and fewer pairs within reach
    assert on[-1] < 0.9 * on[0]                       # it left because pairs were being culled, not on the first dense E-step
    params = dict(rot=np.identity(3), t=np.zeros(3), scale=1.0)
    sigma2 = co.squared_kernel_sum_closed_form(src, tgt)
    for _ in range(k):
        es = co.EstepResult(*cpd_c.expectation_step(co.transform("rigid", params, src), tgt, sigma2, 0.0))
        params, sigma2, q = co.mstep_rigid(src, tgt, es)
    assert rel_err(res.transformation.rot, params["rot"]) < TOL_TF
    assert np.max(np.abs(res.transformation.t - params["t"])) < TOL_TF
    assert abs(res.transformation.scale - params["scale"]) < TOL_TF * params["scale"]
    assert abs(res.sigma2 - sigma2) <= TOL_SIGMA2 * sigma2


@pytest.mark.parametrize("n,m,w", [(40000, 40000, 0.0), (30011, 45007, 0.1)])
def test_stream_mode_equals_grid_mode_and_the_oracle(n, m, w):
    """Dense regime: the matrix-core sweeps cut in stream mode (equal runs of (block, chunk) units, several work items per
    workgroup, a block's partial sums spread over the planes of the workgroups that touched it) against the
    grid of (block, segment) workgroups from the SAME state, and both against the fp64 oracle (cpd.py:71-88).  Both cuts
    evaluate every pair."""
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import synthetic

    src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=29)
    reg, plan = _plan_with_state(src, tgt, 1, 1)
    state = plan.get_params()
    out = {}
    for stream in (True, False):
        plan.set_stream_mode(stream)
        plan.set_dense_engine(1)       # (resets the switch's memory: the next E-steps start in the dense regime)
        plan.set_params(state)
        plan.estep(w)                  # first E-step after the reset takes the matrix cores and counts
        plan.set_params(state)
        plan.estep(w)
        assert plan.last_estep_engines() == (1, 1) and plan.last_estep_lean() == 1
        assert plan.pair_counts() == (float(n) * m, float(n) * m)
        out[stream] = (plan.get_moments(), plan.get_estep())
    res = reg._result_from_params(state)
    ts = co.transform("rigid", dict(rot=res.transformation.rot, t=res.transformation.t, scale=res.transformation.scale), src)
    pt1, p1, px, n_p = cpd_c.expectation_step(ts, tgt, res.sigma2, w)
    for stream in (True, False):
        mom, (g_pt1, g_p1, g_px) = out[stream]
        g_px = g_px + np.outer(g_p1, reg._cx)
        assert abs(mom[0] - n_p) < 2e-6 * n_p, stream
        assert np.max(np.abs(g_pt1 - pt1)) < 2e-5, stream
        assert np.max(np.abs(g_p1 - p1)) < 2e-5 * max(1.0, p1.max()), stream
        assert np.max(np.abs(g_px - px)) < 2e-5 * max(1.0, np.abs(px).max()), stream
    a, b = out[True][0][:23], out[False][0][:23]
    assert np.max(np.abs(a - b)) < 1e-6 * n_p
    # the column side does not depend on how the launch is cut beyond the order fp32 partials are added in
    assert np.max(np.abs(out[True][1][0] - out[False][1][0])) < 1e-6


@pytest.mark.parametrize("kind,single", [("rigid", True), ("rigid", False), ("affine", False)])
def test_fine_grid_of_culling_matrix_core_launches_matches_the_oracle(kind, single):
    """[r5] DESIGN.md 3.1c: few owned blocks against a long streamed cloud (a shard's shape: 9 000 targets = 18 column blocks against
    60 000 sources = 235 chunks) - once the masks skip a tenth of the pairs the matrix-core launches are cut into >= 3 rounds of
    shorter segments (58 column planes instead of 32 here).  Both sweeps pinned to the matrix cores so that they keep running,
    culling, on that grid - as the fused single sweep and as column pass + row pass: E-step moments against the C oracle from the
    same state, and the M-step that follows."""
    from oracle import cpd_c, cpd_numpy as co
    from probreg_amd import _lib, cpd, synthetic

    n, m = 9000, 60000
    if kind == "rigid":
        src, tgt, _ = synthetic.rigid_pair(n, m=m, seed=61)
        reg, kind_id = cpd.RigidCPD(src), _lib.PRG_TF_RIGID
    else:
        src, tgt, _ = synthetic.affine_pair(n, m=m, seed=61)
        reg, kind_id = cpd.AffineCPD(src), _lib.PRG_TF_AFFINE
    assert src.shape[0] == m and tgt.shape[0] == n
    reg._initialize(tgt)
    plan = reg._plan
    plan.set_dense_engine(2)
    plan.set_fused_factor(1e30)
    plan.set_lean_factor(1e30)
    plan.set_moments_only(1 if single else 2)
    culled = 0
    for it in range(14):
        st = reg._result_from_params(plan.get_params())
        plan.estep(0.0)
        assert plan.last_estep_engine() == 1 and plan.last_estep_fused() == (1 if single else 0)
        col_pairs, _row_pairs = plan.pair_counts()
        prev_culled, culled = culled, culled + (col_pairs < 0.9 * float(n) * m)
        plan.mstep(kind_id, True)
        if prev_culled >= 2 and it % 2 == 1:   # (the grid of THIS E-step was chosen from the count of the one before the previous)
            out = reg._result_from_params(plan.get_params())
            tr = st.transformation
            p0 = dict(rot=tr.rot, t=tr.t, scale=float(tr.scale)) if kind == "rigid" else dict(b=tr.b, t=tr.t)
            es = co.EstepResult(*cpd_c.expectation_step(co.transform(kind, p0, src), tgt, st.sigma2, 0.0))
            p, s2, _q = (co.mstep_rigid if kind == "rigid" else co.mstep_affine)(src, tgt, es)
            assert abs(out.sigma2 - s2) <= TOL_SIGMA2 * s2, (it, out.sigma2, s2)
            lin, want = (out.transformation.rot, p["rot"]) if kind == "rigid" else (out.transformation.b, p["b"])
            assert rel_err(lin, want) < TOL_TF and np.max(np.abs(out.transformation.t - p["t"])) < TOL_TF
    assert culled >= 4, culled   # the masks were at work for several iterations: the fine grid ran
