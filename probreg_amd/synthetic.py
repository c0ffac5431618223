"""Seeded synthetic clouds for the benchmark configurations of BASELINE.json.

Isotropic blobs are useless for registration (rotationally symmetric), so the clouds are samples
of an asymmetric closed surface; every config draws source and target as *independent* samples so
that there is no exact point-to-point correspondence.
"""
import numpy as np


def surface(n, seed):
    """n points on an asymmetric tube-like surface, float64 (n, 3)."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(0.0, 2.0 * np.pi, n)
    v = rng.uniform(-1.0, 1.0, n)
    x = (1.0 + 0.3 * np.cos(3.0 * u)) * np.cos(u) * (1.0 - 0.3 * v * v)
    y = 0.6 * (1.0 + 0.2 * np.sin(2.0 * u)) * np.sin(u)
    z = 0.4 * v + 0.15 * np.sin(2.0 * u + v)
    return np.stack([x, y, z], axis=1)


def surface_with_normals(n, seed):
    """Same surface as :func:`surface` plus its unit normals (cross product of the analytic tangents)."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(0.0, 2.0 * np.pi, n)
    v = rng.uniform(-1.0, 1.0, n)
    a, b = 1.0 + 0.3 * np.cos(3.0 * u), 1.0 - 0.3 * v * v
    pts = np.stack([a * np.cos(u) * b, 0.6 * (1.0 + 0.2 * np.sin(2.0 * u)) * np.sin(u), 0.4 * v + 0.15 * np.sin(2.0 * u + v)],
                   axis=1)
    du = np.stack([(-0.9 * np.sin(3.0 * u) * np.cos(u) - a * np.sin(u)) * b,
                   0.6 * (0.4 * np.cos(2.0 * u) * np.sin(u) + (1.0 + 0.2 * np.sin(2.0 * u)) * np.cos(u)),
                   0.3 * np.cos(2.0 * u + v)], axis=1)
    dv = np.stack([a * np.cos(u) * (-0.6 * v), np.zeros(n), 0.4 + 0.15 * np.cos(2.0 * u + v)], axis=1)
    nrm = np.cross(du, dv)
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    return pts, nrm


def rot_zx(deg_z, deg_x):
    a, b = np.deg2rad(deg_z), np.deg2rad(deg_x)
    rz = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    rx = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(b), -np.sin(b)], [0.0, np.sin(b), np.cos(b)]])
    return rz @ rx


def rigid_pair(n, m=None, noise=0.005, seed=0):
    """C1: source = surface(seed), target = independent sample moved by R_z(30)R_x(10), t, + noise."""
    m = n if m is None else m
    src = surface(m, seed)
    tgt = surface(n, seed + 1)
    r = rot_zx(30.0, 10.0)
    t = np.array([0.1, -0.05, 0.02])
    rng = np.random.default_rng(seed + 2)
    tgt = tgt @ r.T + t + rng.normal(0.0, noise, tgt.shape)
    # both paths see the same float32-representable inputs
    return src.astype(np.float32).astype(np.float64), tgt.astype(np.float32).astype(np.float64), (r, t, 1.0)


def affine_pair(n, m=None, noise=0.005, seed=0):
    """C2: target = A x + t with A = R diag(1.1, 0.95, 1.0) + 0.05 shear."""
    m = n if m is None else m
    src = surface(m, seed)
    tgt = surface(n, seed + 1)
    a = rot_zx(20.0, 5.0) @ np.diag([1.1, 0.95, 1.0])
    a[0, 1] += 0.05
    t = np.array([0.05, 0.02, -0.03])
    rng = np.random.default_rng(seed + 2)
    tgt = tgt @ a.T + t + rng.normal(0.0, noise, tgt.shape)
    return src.astype(np.float32).astype(np.float64), tgt.astype(np.float32).astype(np.float64), (a, t)


def nonrigid_pair(n, m=None, noise=0.003, seed=0):
    """C3: target = independent sample displaced by a smooth 0.05*sin(3x) field."""
    m = n if m is None else m
    src = surface(m, seed)
    tgt = surface(n, seed + 1)
    disp = 0.05 * np.sin(3.0 * tgt[:, [1, 2, 0]])
    rng = np.random.default_rng(seed + 2)
    tgt = tgt + disp + rng.normal(0.0, noise, tgt.shape)
    return src.astype(np.float32).astype(np.float64), tgt.astype(np.float32).astype(np.float64)


def pt2pl_pair(n, m=None, noise=0.003, seed=0):
    """Point-to-plane FilterReg case: rigidly moved surface sample with its (moved) analytic normals."""
    m = n if m is None else m
    src = surface(m, seed)
    tgt, nrm = surface_with_normals(n, seed + 1)
    r = rot_zx(12.0, -6.0)
    t = np.array([0.04, 0.03, -0.02])
    rng = np.random.default_rng(seed + 2)
    tgt = tgt @ r.T + t + rng.normal(0.0, noise, tgt.shape)
    nrm = nrm @ r.T
    f = lambda a: a.astype(np.float32).astype(np.float64)
    return f(src), f(tgt), f(nrm), (r, t)


def filterreg_pair(n, m=None, outlier_frac=0.05, noise=0.005, seed=0):
    """C4: rigidly moved surface sample with ``outlier_frac`` uniform outliers in 1.5x the bounding box."""
    m = n if m is None else m
    src = surface(m, seed)
    n_out = int(round(n * outlier_frac))
    tgt = surface(n - n_out, seed + 1)
    r = rot_zx(15.0, 5.0)
    t = np.array([0.05, -0.02, 0.01])
    rng = np.random.default_rng(seed + 2)
    tgt = tgt @ r.T + t + rng.normal(0.0, noise, tgt.shape)
    lo, hi = tgt.min(axis=0), tgt.max(axis=0)
    c, half = 0.5 * (lo + hi), 0.75 * (hi - lo)
    out = rng.uniform(c - half, c + half, (n_out, 3))
    tgt = np.concatenate([tgt, out], axis=0)
    rng.shuffle(tgt, axis=0)
    return src.astype(np.float32).astype(np.float64), tgt.astype(np.float32).astype(np.float64), (r, t)
