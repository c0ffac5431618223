"""The order CPD plans store their clouds in (csrc/morton.h + csrc/spatial_order.hip; no reference counterpart -
probreg keeps the caller's order): the in-order walk of a left-aligned kd-tree with 32-point leaves.  CPU: the host build's
properties; GPU: the device build (what prg_cpd_set_source / prg_cpd_set_target run) gives the same cells."""
import numpy as np
import pytest

from probreg_amd import dist, engine, synthetic


def _leaf_boxes(p, size):
    n = p.shape[0] // size * size
    q = p[:n].reshape(-1, size, p.shape[1])
    return q.min(axis=1), q.max(axis=1)


@pytest.mark.parametrize("n,dim", [(10000, 3), (4097, 2), (33, 3), (20, 3)])
def test_host_kd_order_is_a_permutation_with_compact_aligned_runs(n, dim):
    pts = synthetic.surface(n, seed=7)[:, :dim]
    perm = engine.spatial_order(pts, on_device=False)
    assert np.array_equal(np.sort(perm), np.arange(n))
    assert np.array_equal(perm, engine.spatial_order(pts, on_device=False))   # a function of the cloud alone
    if n < 1024:
        return
    kd = pts[perm].astype(np.float64)
    zc = pts[dist.morton_order(pts)].astype(np.float64)
    for size in (32, 128, 512):   # groups, the blocks a wave owns, the blocks of the matrix-core sweeps: every one a cell
        lo, hi = _leaf_boxes(kd, size)
        lo_z, hi_z = _leaf_boxes(zc, size)
        diag = np.sqrt(((hi - lo) ** 2).sum(axis=1))
        diag_z = np.sqrt(((hi_z - lo_z) ** 2).sum(axis=1))
        assert diag.mean() < diag_z.mean(), (size, diag.mean(), diag_z.mean())       # tighter than runs of the Z-curve
        assert diag.max() <= diag_z.max()                                            # and no run split by a jump of the curve
    # the 2^k leaves of an aligned run are the two halves of a cut across ONE axis: their boxes do not overlap along it
    lo, hi = _leaf_boxes(kd, 32)
    sep = 0
    for a in range(0, lo.shape[0] - 1, 2):
        sep += bool(np.any(hi[a] <= lo[a + 1]) or np.any(hi[a + 1] <= lo[a]))
    assert sep >= 0.95 * (lo.shape[0] // 2)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim", [(100000, 3), (12345, 2), (40, 3), (31, 3)])
def test_device_kd_order_has_the_cells_of_the_host_build(n, dim):
    rng = np.random.default_rng(3)
    pts = (synthetic.surface(n, seed=11)[:, :dim] + 1e-3 * rng.standard_normal((n, dim))).astype(np.float32)  # (no ties at the cuts)
    host = engine.spatial_order(pts, on_device=False)
    dev = engine.spatial_order(pts, on_device=True)
    assert np.array_equal(np.sort(dev), np.arange(n))
    leaves = (n + 31) // 32
    same = 0
    for a in range(leaves):
        same += set(host[32 * a:32 * a + 32].tolist()) == set(dev[32 * a:32 * a + 32].tolist())
    assert same == leaves, (same, leaves)
