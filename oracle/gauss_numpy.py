"""TEST INFRASTRUCTURE - numpy fp64 restatement of the reference's direct Gauss transform and L2 distance.

``gauss_transform_direct`` follows probreg/gauss_transform.py:10-16 (``sum_j w_j exp(-|t_i - s_j|^2 / h^2)``, evaluated
in row blocks of the target instead of ``np.apply_along_axis``), ``GaussTransform.compute``'s weight handling follows
gauss_transform.py:46-60, ``compute_l2_dist`` follows probreg/cost_functions.py:33-41.  Pinned to the reference's own
outputs (tests/golden/gauss_golden.npz, produced by tests/golden/make_golden.py gauss through oracle/ref_import.py) by
tests/test_oracle_gauss.py.  Never imported by probreg_amd/.
"""
import numpy as np


def gauss_transform_direct(source, target, weights, h, block=512):
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    h2 = h * h
    out = np.empty(target.shape[0])
    for i0 in range(0, target.shape[0], block):
        t = target[i0:i0 + block]
        d2 = np.sum(np.square(t[:, None, :] - source[None, :, :]), axis=2)  # gauss_transform.py:15
        out[i0:i0 + block] = np.exp(-d2 / h2) @ weights
    return out


def compute(source, h, target, weights=None):
    """GaussTransform(source, h).compute(target, weights) (gauss_transform.py:46-60)."""
    if weights is None:
        weights = np.ones(np.asarray(source).shape[0])
    weights = np.asarray(weights)
    if weights.ndim == 1:
        return gauss_transform_direct(source, target, weights, h)
    if weights.ndim == 2:
        return np.r_[[gauss_transform_direct(source, target, w, h) for w in weights]]
    raise ValueError("weights.ndim must be 1 or 2.")


def compute_l2_dist(mu_source, phi_source, mu_target, phi_target, sigma):
    """cost_functions.py:33-41."""
    z = np.power(2.0 * np.pi * sigma ** 2, mu_source.shape[1] * 0.5)
    h = np.sqrt(2.0) * sigma
    phi_j_e = compute(mu_target, h, mu_source, phi_target / z)
    phi_mu_j_e = compute(mu_target, h, mu_source, phi_target * mu_target.T / z).T
    g = (phi_source * phi_j_e * mu_source.T - phi_source * phi_mu_j_e.T).T / (2.0 * sigma ** 2)
    return -np.dot(phi_source, phi_j_e), g
