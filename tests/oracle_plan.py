"""An oracle-backed stand-in for ``probreg_amd.engine.CpdPlan`` (TEST INFRASTRUCTURE).

On a box without a GPU the product's host logic - ``probreg_amd.cpd``'s drivers, the fp64 centring, the parameter block
conventions, ``probreg_amd.dist``'s spatial sharding and the per-iteration all-reduce - can still run if something
produces the MOMENTS block a plan would.  This class does that with the numpy oracle, honouring the plan's contract
(include/probreg_hip.h: MOMENTS / PARAMS layouts, init_sums -> all-reduce -> init_params, estep -> all-reduce -> mstep).
Tests monkeypatch it in for ``CpdPlan``; it is never imported by the product.
"""
import numpy as np
import torch

from oracle import cpd_numpy as co

NMOM = NPAR = 32


class OraclePlan(object):
    def __init__(self, device=None, stream=None):
        self.device = 0
        self._moments_tensor = None
        self.mom = np.zeros(NMOM)
        self.par = np.zeros(NPAR)
        self.m = self.n = self.dim = 0

    # -- uploads ---------------------------------------------------------------------------------
    def set_options(self, **kw):
        pass

    def set_source(self, source):
        self.src = np.asarray(source, dtype=np.float32).astype(np.float64)  # the plan stores float32 clouds
        self.m, self.dim = self.src.shape

    def set_target(self, target_local, n_global=None):
        self.tgt = np.asarray(target_local, dtype=np.float32).astype(np.float64)
        self.n = self.tgt.shape[0]
        self.n_global = int(n_global or self.n)

    def moments_tensor(self):
        if self._moments_tensor is None:
            self._moments_tensor = torch.zeros(NMOM, dtype=torch.float64)
            self.mom = self._moments_tensor.numpy()  # shared memory: the all-reduce happens in place
        return self._moments_tensor

    # -- EM pieces ---------------------------------------------------------------------------------
    def init_sums(self):
        self.mom[:] = 0.0
        self.mom[24:24 + self.dim] = self.tgt.sum(axis=0)
        self.mom[27] = float(np.sum(self.tgt * self.tgt))

    def init_params(self, init16=None):
        m, n, d = float(self.m), float(self.n_global), self.dim
        ts, ssum, s2 = self.mom[24:27], np.zeros(3), float(np.sum(self.src * self.src))
        ssum[:d] = self.src.sum(axis=0)
        total = m * self.mom[27] + n * s2 - 2.0 * float(ts @ ssum)
        self.par[:] = 0.0
        if init16 is not None:
            init16 = np.asarray(init16, dtype=np.float64)
            dl = init16[13:16]
            total += 2.0 * float(dl @ (m * ts - n * ssum)) + m * n * float(dl @ dl)
            self.par[:13] = init16[:13]
        else:
            self.par[[0, 4, 8, 12]] = 1.0
        sigma2 = total / (d * m * n)
        self.par[13] = sigma2
        self.par[14] = 1.0 + n * d * 0.5 * np.log(sigma2)
        self.mom[24:] = 0.0

    def estep(self, w=0.0):
        d = self.dim
        lin, t, s = self.par[:9].reshape(3, 3)[:d, :d], self.par[9:9 + d], self.par[12]
        z = s * self.src @ lin.T + t
        sigma2 = self.par[13]
        dd = ((z[:, None, :] - self.tgt[None, :, :]) ** 2).sum(axis=2)
        k = np.exp(dd * (-1.0 / (2.0 * sigma2)))
        c = (2.0 * np.pi * sigma2) ** (d * 0.5) * w / (1.0 - w) * self.m / self.n_global  # GLOBAL N (cpd.py:78-79)
        den = k.sum(axis=0)
        den[den == 0] = co.EPS32
        den += c
        p = k / den
        es = co.EstepResult(p.sum(axis=0), p.sum(axis=1), p @ self.tgt, float(p.sum()))
        self.mom[:] = co.moments_from_estep(self.src, self.tgt, es)
        self.last_estep = es

    def mstep(self, kind, update_scale=True):
        d = self.dim
        res = co.mstep_from_moments("rigid" if kind == 0 else "affine", self.mom, d, bool(update_scale))
        lin = np.identity(3)
        lin[:d, :d] = res.params["rot"] if kind == 0 else res.params["b"]
        self.par[:9] = lin.ravel()
        self.par[9:12] = 0.0
        self.par[9:9 + d] = res.params["t"]
        self.par[12] = res.params.get("scale", 1.0)
        self.par[13], self.par[14], self.par[15] = res.sigma2, res.q, self.mom[0]
        self.par[16] += 1.0

    def get_params(self):
        return self.par.copy()

    def set_params(self, p):
        self.par[:] = np.asarray(p, dtype=np.float64)

    def get_moments(self):
        return self.mom.copy()

    def close(self):
        pass
