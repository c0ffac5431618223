"""Host-side logic that needs no GPU: sharding, parameter-block algebra, API surface."""
import inspect

import numpy as np
import pytest

from probreg_amd import cpd, dist, synthetic, transformation as tf


def test_shard_bounds_cover_and_balance():
    for n in (1, 7, 8, 100000, 200001):
        for world in (1, 2, 3, 8):
            if n < world:
                continue
            spans = [dist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans[:-1], spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dist.shard_bounds(10, 3, 2)


def test_world_without_init():
    assert dist.world() == (0, 1)


def test_params_block_and_centring_roundtrip():
    rng = np.random.default_rng(1)
    rot = synthetic.rot_zx(25.0, -10.0)
    t = rng.normal(size=3)
    s = 1.3
    cy, cx = rng.normal(size=3), rng.normal(size=3)
    y = rng.normal(size=(50, 3))
    # the centred-frame translation used by RigidCPD._initialize / _result_from_params
    t_c = t + s * rot @ cy - cx
    z_centred = s * (y - cy) @ rot.T + t_c
    z = s * y @ rot.T + t
    assert np.allclose(z_centred + cx, z)
    blk = cpd._params_block(rot[:2, :2], t[:2], s, 2)
    assert blk.shape == (16,) and blk[8] == 1.0 and blk[12] == s and blk[11] == 0.0


def test_api_surface_matches_reference_names():
    sig = inspect.signature(cpd.registration_cpd)
    assert list(sig.parameters)[:8] == ["source", "target", "tf_type_name", "w", "maxiter", "tol", "callbacks",
                                        "use_cuda"]
    assert sig.parameters["maxiter"].default == 50 and sig.parameters["tol"].default == 0.001
    assert cpd.EstepResult._fields == ("pt1", "p1", "px", "n_p")
    assert cpd.MstepResult._fields == ("transformation", "sigma2", "q")
    for cls in (cpd.RigidCPD, cpd.AffineCPD, cpd.NonRigidCPD):
        for meth in ("set_source", "set_callbacks", "expectation_step", "maximization_step", "registration"):
            assert hasattr(cls, meth)
    assert list(inspect.signature(cpd.RigidCPD.__init__).parameters)[1:5] == ["source", "update_scale",
                                                                             "tf_init_params", "use_cuda"]
    assert list(inspect.signature(cpd.NonRigidCPD.__init__).parameters)[1:5] == ["source", "beta", "lmd", "use_cuda"]


def test_maximization_steps_are_static_with_the_reference_signatures():
    """cpd.py:160-162, 219-221, 284-292, 377-388: every ``_maximization_step`` is a @staticmethod."""
    want = {
        cpd.RigidCPD: ["source", "target", "estep_res", "sigma2_p", "update_scale", "xp"],
        cpd.AffineCPD: ["source", "target", "estep_res", "sigma2_p", "xp"],
        cpd.NonRigidCPD: ["source", "target", "estep_res", "sigma2_p", "tf_obj", "lmd", "xp"],
        cpd.ConstrainedNonRigidCPD: ["source", "target", "estep_res", "sigma2_p", "tf_obj", "lmd", "alpha", "p1_tilde",
                                     "px_tilde", "xp"],
    }
    for cls, names in want.items():
        assert isinstance(inspect.getattr_static(cls, "_maximization_step"), staticmethod), cls
        assert list(inspect.signature(cls._maximization_step).parameters)[:len(names)] == names, cls


def test_unknown_type_raises_value_error():
    x = np.zeros((4, 3))
    with pytest.raises(ValueError):
        cpd.registration_cpd(x, x, "projective")


def test_transformations_are_plain_value_objects():
    rot = synthetic.rot_zx(30.0, 0.0)
    r = tf.RigidTransformation(rot, np.array([1.0, 2.0, 3.0]), 2.0)
    p = np.array([[1.0, 0.0, 0.0]])
    assert np.allclose(r.transform(p), 2.0 * p @ rot.T + [1.0, 2.0, 3.0])
    inv = r.inverse()
    assert np.allclose(inv.transform(r.transform(p)), p)
    a = tf.AffineTransformation(np.diag([1.0, 2.0, 3.0]), np.ones(3))
    assert np.allclose(a.transform(p), [[2.0, 1.0, 1.0]])


def test_synthetic_generators_are_seeded():
    a, b, _ = synthetic.rigid_pair(100, seed=3)
    a2, b2, _ = synthetic.rigid_pair(100, seed=3)
    assert np.array_equal(a, a2) and np.array_equal(b, b2)
    assert a.shape == (100, 3) and np.array_equal(a, a.astype(np.float32).astype(np.float64))
    s, t, _ = synthetic.filterreg_pair(1000, seed=1)
    assert t.shape == (1000, 3)


def test_spatial_shards_partition_the_target_into_compact_patches():
    """Every rank derives the same Morton order: the shards are a partition, near-equal in size, and each one is
    spatially compact (its bounding box is a fraction of the cloud's) - what keeps the culled sweeps effective."""
    from probreg_amd import synthetic

    _, tgt, _ = synthetic.rigid_pair(4001, m=10, seed=3)
    world = 8
    shards = [dist.spatial_shard(tgt, r, world) for r in range(world)]
    allrows = np.sort(np.concatenate(shards))
    assert np.array_equal(allrows, np.arange(tgt.shape[0]))
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    whole = np.prod(tgt.max(axis=0) - tgt.min(axis=0))
    vols = [np.prod(tgt[s].max(axis=0) - tgt[s].min(axis=0)) for s in shards]
    assert np.median(vols) < 0.3 * whole
    # one rank: the caller's order is kept (the plan sorts on upload)
    assert np.array_equal(dist.spatial_shard(tgt, 0, 1), np.arange(tgt.shape[0]))
    # the order is a pure function of the coordinates
    assert np.array_equal(dist.morton_order(tgt), dist.morton_order(tgt.copy()))
    assert np.array_equal(dist.morton_order(tgt[:, :2]), dist.morton_order(tgt[:, :2].copy()))


def test_engine_switch_bounds_follow_the_cost_model():
    """prg_cpd_engine_bounds: below how many evaluated pairs per owned point the matrix-core sweeps are left.
    Host arithmetic - checked here against the crossovers measured on MI355X (profiles/r3_engine_switch_*.log, r4_engine_switch_*.log): within one
    EM iteration, i.e. well within a factor 1.6 in pairs, of where the two engines actually cross."""
    from probreg_amd import engine

    # row pass: round 4's re-measurement (lean matrix-core row pass against vector-pipe sweeps that skip at 2^-48,
    # profiles/r4_engine_switch_*.log); column pass: where the two engines crossed in round 3's logs - round 4's own switch
    # (same constants) still leaves within one EM iteration of the faster engine in every one of these configurations
    measured = {  # (M, N_local): (column-pass crossover, row-pass crossover) in evaluated pairs per owned point
        (30000, 30000): (15000.0, 18600.0),
        (50000, 50000): (13000.0, 13400.0),
        (100000, 100000): (15300.0, 18000.0),
        (250000, 250000): (21000.0, 37500.0),
        (100000, 50000): (None, 12400.0),      # rank 0 of 2
        (100000, 25000): (27000.0, 6300.0),    # rank 0 of 4
        (100000, 12500): (63000.0, 5300.0),    # rank 0 of 8
    }
    for (m, n_local), (col, row) in measured.items():
        c, r = engine.engine_bounds(m, n_local)
        assert col is None or col / 1.6 < c < col * 1.6, (m, n_local, c, col)
        assert row / 1.6 < r < row * 1.6, (m, n_local, r, row)
    # a shard's column pass has too few workgroups to fill the chip: it leaves the matrix cores earlier (at MORE pairs per target)
    c1, _ = engine.engine_bounds(100000, 100000)
    c4, _ = engine.engine_bounds(100000, 25000)
    c8, _ = engine.engine_bounds(100000, 12500)
    assert c1 < c4 < c8
    # ... while its row pass streams a short local target: never more than that target holds
    for n_local in (12500, 25000, 50000):
        _, r = engine.engine_bounds(100000, n_local)
        assert r < n_local
    with pytest.raises(Exception):
        engine.engine_bounds(0, 100)


def test_native_comm_decision_is_not_cached_without_a_process_group(monkeypatch):
    """A single-process registration BEFORE init_process_group must leave nothing behind: a cached "no communicator" would
    keep this rank out of the collective set-up its peers enter once the group exists (mismatched collectives)."""
    from probreg_amd import dist

    monkeypatch.delenv("PROBREG_NATIVE_RCCL", raising=False)
    dist._native.clear()
    assert dist.native_comm(0) is None
    assert dist._native == {}
    # ... and a plan that held a communicator is detached when the communicator goes away
    class _Plan(object):
        _h = 1

        def set_comm(self, comm):
            self._comm = comm

    comm = object.__new__(dist.NativeComm)
    comm._h, comm._plans = None, __import__("weakref").WeakSet()
    plan = _Plan()
    plan._comm = comm
    comm._attached(plan)

    class _Lib(object):
        class lib(object):
            destroyed = []

            @staticmethod
            def prg_comm_destroy(h):
                _Lib.lib.destroyed.append(h)

    comm._lib, comm._h = _Lib, 7
    comm.close()
    assert plan._comm is None and _Lib.lib.destroyed == [7] and comm._h is None
