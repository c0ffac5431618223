"""L2 distance between Gaussian mixtures on the GPU (reference probreg/cost_functions.py:33-41).

Only ``compute_l2_dist`` - the Gauss-transform-bound kernel that the reference's SVR / GMMReg cost functions
call every BFGS evaluation - is provided; the optimiser drivers themselves
(`l2dist_regs.py`, scikit-learn feature fitting) are out of scope.  The reference evaluates it through IFGT for
wide kernels; here it is always the exact direct transform.
"""
import numpy as np

from . import gauss_transform as gt


def compute_l2_dist(mu_source, phi_source, mu_target, phi_target, sigma):
    """Returns (-sum_ij phi_s_i phi_t_j N(mu_s_i - mu_t_j; 2 sigma^2), gradient w.r.t. mu_source)."""
    mu_source = np.asarray(mu_source, dtype=np.float64)
    mu_target = np.asarray(mu_target, dtype=np.float64)
    phi_source = np.asarray(phi_source, dtype=np.float64)
    phi_target = np.asarray(phi_target, dtype=np.float64)
    z = np.power(2.0 * np.pi * sigma ** 2, mu_source.shape[1] * 0.5)
    gtrans = gt.GaussTransform(mu_target, np.sqrt(2.0) * sigma)
    # one launch group for the scalar weights and the D coordinate-weighted rows (cost_functions.py:38-39)
    weights = np.concatenate([(phi_target / z)[None, :], phi_target * mu_target.T / z], axis=0)
    res = gtrans.compute(mu_source, weights)
    phi_j_e = res[0]
    phi_mu_j_e = res[1:].T
    g = (phi_source * phi_j_e * mu_source.T - phi_source * phi_mu_j_e.T).T / (2.0 * sigma ** 2)
    return -np.dot(phi_source, phi_j_e), g
