"""Thin object wrapper over the ``prg_cpd_*`` C ABI (one plan = one GPU = one HIP stream).

Device memory for the clouds and workspaces is owned by the library; PyTorch is used only to
pick the device / stream of the current process and to give ``torch.distributed`` (RCCL) a
tensor view of the 32-double moment block that is all-reduced once per EM iteration.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib, ptr


def _current_device_and_stream(device=None):
    """(device index, hipStream_t as int) of the calling process; torch is optional plumbing."""
    try:
        import torch

        if torch.cuda.is_available():
            dev = torch.cuda.current_device() if device is None else int(device)
            return dev, int(torch.cuda.current_stream(dev).cuda_stream)
    except ImportError:  # pragma: no cover
        pass
    return (0 if device is None else int(device)), 0


def spatial_order(points, on_device=True):
    """The order a CPD plan stores ``points`` (n, 2|3) in: sorted position -> original index (prg_spatial_order; the kd-tree
    order of DESIGN.md 3.1b, built on the GPU as the plans do, or on the host)."""
    import numpy as np

    pts = np.ascontiguousarray(points, dtype=np.float32)
    perm = np.empty(pts.shape[0], dtype=np.int32)
    check(lib.prg_spatial_order(pts.ctypes.data_as(ctypes.c_void_p), int(pts.shape[0]), int(pts.shape[1]), 1 if on_device else 0,
                                perm.ctypes.data_as(ctypes.c_void_p)))
    return perm


def engine_bounds(m, n_local):
    """(column bound, row bound) of the dense-regime engine switch for a source of ``m`` points and a local target of
    ``n_local``: evaluated pairs per owned point below which the matrix-core sweep is left (prg_cpd_engine_bounds; host
    arithmetic, no GPU needed)."""
    a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
    check(lib.prg_cpd_engine_bounds(int(m), int(n_local), ctypes.byref(a), ctypes.byref(b)))
    return float(a.value), float(b.value)


class CpdPlan(object):
    """Owns one ``prg_cpd`` handle."""

    def __init__(self, device=None, stream=None):
        _lib.require_gpu()
        dev, st = _current_device_and_stream(device)
        if stream is not None:
            st = int(stream)
        self.device = dev
        self.stream = st
        self._h = ctypes.c_void_p()
        check(lib.prg_cpd_create(ctypes.byref(self._h), dev, ctypes.c_void_p(st)))
        self._moments_tensor = None
        self._comm = None
        self.m = self.n = self.dim = 0

    # -- life cycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.prg_cpd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    # -- uploads ----------------------------------------------------------------------------
    @staticmethod
    def _f32(a):
        if hasattr(a, "data_ptr"):  # torch tensor, host or device
            import torch

            assert a.dtype == torch.float32 and a.is_contiguous()
            return a
        return np.ascontiguousarray(a, dtype=np.float32)

    def set_options(self, sort_source=True, sort_target=True, cull=True):
        check(lib.prg_cpd_set_options(self._h, int(sort_source), int(sort_target), int(cull)))

    def set_dense_engine(self, mode=1, bound=0.0):
        """0: vector-pipe sweeps only, 1: matrix-core sweeps while they are the faster engine (default; decided on the device
        from the pairs the previous E-step evaluated), 2: both sweeps on the matrix cores always.  ``bound`` > 0: leave the
        matrix-core column pass below that many evaluated source points per target (prg_cpd_set_dense_engine)."""
        check(lib.prg_cpd_set_dense_engine(self._h, int(mode), float(bound)))

    def last_estep_engine(self):
        """1 if the last E-step ran on the matrix cores, 0 for the vector-pipe sweeps."""
        e = ctypes.c_int(0)
        check(lib.prg_cpd_last_estep_engine(self._h, ctypes.byref(e)))
        return int(e.value)

    def set_sparse_engine(self, mode=1):
        """1: sparse-regime sweeps over the device-built work queue when both clouds are large (default), 2: always,
        0: never (the grid-per-(block, segment) culled sweeps)."""
        check(lib.prg_cpd_set_sparse_engine(self._h, int(mode)))

    def last_estep_engines(self):
        """(column pass, row pass) of the last E-step: 1 = matrix cores, 0 = vector-pipe sweeps."""
        c, r = ctypes.c_int(0), ctypes.c_int(0)
        check(lib.prg_cpd_last_estep_engines(self._h, ctypes.byref(c), ctypes.byref(r)))
        return int(c.value), int(r.value)

    def last_estep_lean(self):
        """1 if the last E-step's matrix-core row pass ran without its residual sums (prg_cpd_last_estep_lean)."""
        v = ctypes.c_int(0)
        check(lib.prg_cpd_last_estep_lean(self._h, ctypes.byref(v)))
        return int(v.value)

    def set_stream_mode(self, on=True):
        """Dense-regime matrix-core launches cut into equal runs of units over the chip's workgroup slots (default) or into the
        grid of (block, segment) workgroups (prg_cpd_set_stream_mode)."""
        check(lib.prg_cpd_set_stream_mode(self._h, 1 if on else 0))

    def set_lean_factor(self, factor=-1.0):
        """Lean matrix-core row pass while mean |x|^2 / (sigma2 D) <= factor (default 64; 0 never; < 0 restores the default)."""
        check(lib.prg_cpd_set_lean_factor(self._h, float(factor)))

    def set_source(self, source):
        a = self._f32(source)
        self.m, self.dim = int(a.shape[0]), int(a.shape[1])
        check(lib.prg_cpd_set_source(self._h, ptr(a), self.m, self.dim))

    def set_target(self, target_local, n_global=None):
        a = self._f32(target_local)
        self.n = int(a.shape[0])
        check(lib.prg_cpd_set_target(self._h, ptr(a), self.n, int(a.shape[1]), int(n_global or self.n)))

    def set_tuning(self, r_col=0, seg_col=0, r_row=0, seg_row=0):
        check(lib.prg_cpd_set_tuning(self._h, r_col, seg_col, r_row, seg_row))

    # -- multi-GPU: the library's own RCCL all-reduce on the plan's stream -------------------
    def set_comm(self, comm):
        """Attach a ``dist.NativeComm`` (None detaches): init_sums / estep then end with the all-reduce themselves."""
        check(lib.prg_cpd_set_comm(self._h, comm._h if comm is not None else None))
        old = getattr(self, "_comm", None)
        if old is not None and old is not comm:
            old._detached(self)
        self._comm = comm  # keeps the communicator alive as long as the plan uses it
        if comm is not None:
            comm._attached(self)  # ... and the communicator detaches the plan before it is destroyed (NativeComm.close)

    def iterate(self, kind, update_scale, w, n_iter):
        """``n_iter`` EM iterations enqueued back to back inside the library (prg_cpd_iterate)."""
        check(lib.prg_cpd_iterate(self._h, int(kind), 1 if update_scale else 0, float(w), int(n_iter)))

    def set_moments_only(self, mode=1):
        """1: every E-step of this plan feeds a rigid M-step only - the dense regime may run the fused single sweep (no p1 / px
        afterwards); 0 (default): only ``iterate`` does that; 2: never (prg_cpd_set_moments_only)."""
        check(lib.prg_cpd_set_moments_only(self._h, int(mode)))

    def set_resid_sweep(self, on=True):
        """Residual-form single sweep where a rigid iteration's column pass runs on the vector pipe (prg_cpd_set_resid_sweep)."""
        check(lib.prg_cpd_set_resid_sweep(self._h, 1 if on else 0))

    def set_fused_factor(self, factor=256.0):
        """Fused single sweep while mean |x|^2 / (sigma2 D) <= factor (prg_cpd_set_fused_factor)."""
        check(lib.prg_cpd_set_fused_factor(self._h, float(factor)))

    def last_estep_fused(self):
        v = ctypes.c_int(0)
        check(lib.prg_cpd_last_estep_fused(self._h, ctypes.byref(v)))
        return int(v.value)

    # -- moment block as a torch tensor (for a caller-side all-reduce: gloo, tests) -----------
    def moments_tensor(self):
        """Allocate (once) a torch fp64 tensor on the plan's device and bind it as MOMENTS."""
        if self._moments_tensor is None:
            import torch

            t = torch.zeros(_lib.PRG_NMOMENTS, dtype=torch.float64, device="cuda:%d" % self.device)
            check(lib.prg_cpd_bind_moments(self._h, ctypes.c_void_p(t.data_ptr())))
            self._moments_tensor = t
        return self._moments_tensor

    def rowacc_tensor_view(self):
        """(device pointer, count) of the per-point fp64 E-step block (non-rigid all-reduce payload)."""
        p = ctypes.c_void_p()
        cnt = ctypes.c_int64()
        check(lib.prg_cpd_rowacc_ptr(self._h, ctypes.byref(p), ctypes.byref(cnt)))
        return p.value, int(cnt.value)

    # -- EM pieces ----------------------------------------------------------------------------
    def init_sums(self):
        check(lib.prg_cpd_init_sums(self._h))

    def init_params(self, init16=None):
        """``init16``: linear part (9), t (3), scale, delta = origin_target - origin_source (3)."""
        if init16 is None:
            check(lib.prg_cpd_init_params(self._h, None))
        else:
            a = np.ascontiguousarray(init16, dtype=np.float64)
            assert a.size == 16
            check(lib.prg_cpd_init_params(self._h, ptr(a)))

    def estep(self, w=0.0):
        check(lib.prg_cpd_estep(self._h, float(w)))

    def estep_timed(self, w=0.0):
        """E-step with per-kernel HIP-event timing: dict of milliseconds (see prg_cpd_estep_timed)."""
        ms = np.zeros(6, dtype=np.float32)
        check(lib.prg_cpd_estep_timed(self._h, float(w), ptr(ms)))
        names = ("transform", "colpass", "colfinal", "rowpass", "moments", "total")
        return dict(zip(names, [float(v) for v in ms]))

    def pair_counts(self):
        """(column-pass pairs, row-pass pairs) the last E-step actually evaluated (prg_cpd_pair_counts)."""
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        check(lib.prg_cpd_pair_counts(self._h, ctypes.byref(a), ctypes.byref(b)))
        return float(a.value), float(b.value)

    def mstep(self, kind, update_scale=True):
        check(lib.prg_cpd_mstep(self._h, int(kind), 1 if update_scale else 0))

    def mstep_nonrigid(self, lmd):
        check(lib.prg_cpd_mstep_nonrigid(self._h, float(lmd)))

    def get_params(self):
        out = np.empty(_lib.PRG_NPARAMS, dtype=np.float64)
        check(lib.prg_cpd_get_params(self._h, ptr(out)))
        return out

    def set_params(self, params):
        a = np.ascontiguousarray(params, dtype=np.float64)
        assert a.size == _lib.PRG_NPARAMS
        check(lib.prg_cpd_set_params(self._h, ptr(a)))

    def get_moments(self):
        out = np.empty(_lib.PRG_NMOMENTS, dtype=np.float64)
        check(lib.prg_cpd_get_moments(self._h, ptr(out)))
        return out

    def get_estep(self):
        pt1 = np.empty(self.n, dtype=np.float64)
        p1 = np.empty(self.m, dtype=np.float64)
        px = np.empty((self.m, self.dim), dtype=np.float64)
        check(lib.prg_cpd_get_estep(self._h, ptr(pt1), ptr(p1), ptr(px)))
        return pt1, p1, px

    def get_estep_pt1(self):
        """Column sums pt1 alone (available after every E-step, the fused single sweep included)."""
        pt1 = np.empty(self.n, dtype=np.float64)
        check(lib.prg_cpd_get_estep(self._h, ptr(pt1), None, None))
        return pt1

    def get_tsource(self):
        out = np.empty((self.m, self.dim), dtype=np.float32)
        check(lib.prg_cpd_get_tsource(self._h, ptr(out)))
        return out

    def moments_from_estep(self, pt1, p1, px):
        pt1 = np.ascontiguousarray(pt1, dtype=np.float64)
        p1 = np.ascontiguousarray(p1, dtype=np.float64)
        px = np.ascontiguousarray(px, dtype=np.float64)
        assert pt1.shape == (self.n,) and p1.shape == (self.m,) and px.shape == (self.m, self.dim)
        check(lib.prg_cpd_moments_from_estep(self._h, ptr(pt1), ptr(p1), ptr(px)))

    # -- non-rigid ----------------------------------------------------------------------------
    def build_g(self, beta):
        check(lib.prg_cpd_nonrigid_build_g(self._h, float(beta)))

    def set_nonrigid_solver(self, mode=1, max_rank=0, tol=0.0):
        """Before build_g: 1 = low-rank factor of G when its rank allows (default), 0 = dense G + M x M Cholesky."""
        check(lib.prg_cpd_nonrigid_set_solver(self._h, int(mode), int(max_rank), float(tol)))

    def nonrigid_rank(self):
        """Rank of the kernel factor the plan holds; 0 when it holds the dense matrix."""
        r = ctypes.c_int(0)
        check(lib.prg_cpd_nonrigid_rank(self._h, ctypes.byref(r)))
        return int(r.value)

    def get_g(self):
        out = np.empty((self.m, self.m), dtype=np.float32)
        check(lib.prg_cpd_nonrigid_get_g(self._h, ptr(out)))
        return out

    def set_w(self, w):
        a = np.ascontiguousarray(w, dtype=np.float64)
        assert a.shape == (self.m, self.dim)
        check(lib.prg_cpd_nonrigid_set_w(self._h, ptr(a)))

    def get_w(self):
        out = np.empty((self.m, self.dim), dtype=np.float64)
        check(lib.prg_cpd_nonrigid_get_w(self._h, ptr(out)))
        return out

    def set_priors(self, p1_tilde, px_tilde, alpha):
        a = np.ascontiguousarray(p1_tilde, dtype=np.float64)
        b = np.ascontiguousarray(px_tilde, dtype=np.float64)
        assert a.shape == (self.m,) and b.shape == (self.m, self.dim)
        check(lib.prg_cpd_nonrigid_set_priors(self._h, ptr(a), ptr(b), float(alpha)))

    def nonrigid_apply(self):
        out = np.empty((self.m, self.dim), dtype=np.float64)
        check(lib.prg_cpd_nonrigid_apply(self._h, ptr(out)))
        return out

    # -- Bayesian CPD --------------------------------------------------------------------------
    def set_source_weights(self, log_weights, uniform_ratio=0.0):
        """ln a_m <= 0 per source point (None clears); ``uniform_ratio`` > 0 replaces M/N in the outlier constant."""
        if log_weights is None:
            check(lib.prg_cpd_set_source_weights(self._h, None, float(uniform_ratio)))
            return
        a = np.ascontiguousarray(log_weights, dtype=np.float64)
        assert a.shape == (self.m,)
        check(lib.prg_cpd_set_source_weights(self._h, ptr(a), float(uniform_ratio)))

    def bcpd_build_g(self, c=1.0):
        check(lib.prg_cpd_bcpd_build_g(self._h, float(c)))

    def bcpd_solve(self, lmd, cfac, resid, nu=None):
        """(v_hat [m x dim], diag(Sigma) [m]) for ``nu`` (default: the p1 of the last E-step) - prg_cpd_bcpd_solve."""
        r = np.ascontiguousarray(resid, dtype=np.float64)
        assert r.shape == (self.m, self.dim)
        v = np.empty((self.m, self.dim), dtype=np.float64)
        d = np.empty(self.m, dtype=np.float64)
        nu_a = None if nu is None else np.ascontiguousarray(nu, dtype=np.float64)
        assert nu_a is None or nu_a.shape == (self.m,)
        check(lib.prg_cpd_bcpd_solve(self._h, float(lmd), float(cfac), None if nu_a is None else ptr(nu_a), ptr(r),
                                     ptr(v), ptr(d)))
        return v, d

    def synchronize(self):
        # a host read-back of the parameter block synchronises the plan's stream
        self.get_params()
