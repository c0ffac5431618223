"""ctypes binding of ``csrc/libprobreg_hip.so`` (the C ABI of include/probreg_hip.h).

The shared library is the product: there is no Python / NumPy fallback for anything it
computes.  If it is missing or fails to load, importing this module raises ``ImportError``
with the build command, and every wrapper raises on a non-zero status.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PROBREG_HIP_LIB: an alternative build of the same library (instrumented / experimental), for tools only
LIB_PATH = os.environ.get("PROBREG_HIP_LIB") or os.path.join(_HERE, "csrc", "libprobreg_hip.so")

PRG_OK = 0
PRG_ERR_INVALID = -1
PRG_ERR_HIP = -2
PRG_ERR_STATE = -3
PRG_ERR_NOMEM = -4

PRG_TF_RIGID = 0
PRG_TF_AFFINE = 1
PRG_TF_NONRIGID = 2

PRG_NMOMENTS = 32
PRG_NPARAMS = 32
PRG_COMM_ID_BYTES = 128


class ProbregHipError(RuntimeError):
    """A libprobreg_hip call failed (HIP runtime error or call-order violation)."""


def _load():
    # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so) and the dynamic loader keeps whichever copy
    # of that soname is loaded FIRST.  If ours came first (the system's /opt/rocm runtime), torch.cuda would later find
    # "no GPUs" in the same process - and with it RCCL; loading torch first makes both sides share torch's runtime,
    # whatever the import order of the caller.  torch stays optional.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch absent, or present with broken shared libraries (OSError): stay usable
        pass
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "probreg_amd: %s not found. Build it with `python __graft_entry__.py build` "
            "(or `make -C probreg_amd/csrc`); there is no CPU fallback." % LIB_PATH
        )
    try:
        return ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the ROCm install
        raise ImportError("probreg_amd: cannot load %s: %s" % (LIB_PATH, e))


lib = _load()

_c = ctypes
_vp, _i, _i64, _d = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_double
_pp = _c.POINTER(_c.c_void_p)

# name -> argtypes; every function returns int except prg_last_error.  This table is the Python
# mirror of include/probreg_hip.h and tests/test_abi.py checks the two against each other.
SIGNATURES = {
    "prg_version": [],
    "prg_device_count": [_c.POINTER(_i)],
    "prg_cpd_create": [_pp, _i, _vp],
    "prg_cpd_destroy": [_vp],
    "prg_spatial_order": [_vp, _i64, _i, _i, _vp],
    "prg_cpd_set_options": [_vp, _i, _i, _i],
    "prg_cpd_set_dense_engine": [_vp, _i, _d],
    "prg_cpd_last_estep_engine": [_vp, _c.POINTER(_i)],
    "prg_cpd_set_sparse_engine": [_vp, _i],
    "prg_cpd_engine_bounds": [_i64, _i64, _c.POINTER(_d), _c.POINTER(_d)],
    "prg_cpd_last_estep_lean": [_vp, _c.POINTER(_i)],
    "prg_cpd_set_lean_factor": [_vp, _d],
    "prg_cpd_set_stream_mode": [_vp, _i],
    "prg_cpd_last_estep_engines": [_vp, _c.POINTER(_i), _c.POINTER(_i)],
    "prg_cpd_set_source": [_vp, _vp, _i64, _i],
    "prg_cpd_set_target": [_vp, _vp, _i64, _i, _i64],
    "prg_cpd_bind_moments": [_vp, _vp],
    "prg_cpd_moments_ptr": [_vp, _pp],
    "prg_cpd_params_ptr": [_vp, _pp],
    "prg_comm_available": [_c.POINTER(_i)],
    "prg_comm_unique_id": [_vp],
    "prg_comm_create": [_pp, _vp, _i, _i, _i],
    "prg_comm_adopt": [_pp, _vp, _i],
    "prg_comm_info": [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i64)],
    "prg_comm_destroy": [_vp],
    "prg_comm_all_reduce_f64": [_vp, _vp, _i64, _vp],
    "prg_cpd_set_comm": [_vp, _vp],
    "prg_cpd_iterate": [_vp, _i, _i, _d, _i],
    "prg_cpd_set_moments_only": [_vp, _i],
    "prg_cpd_last_estep_fused": [_vp, _c.POINTER(_i)],
    "prg_cpd_set_fused_factor": [_vp, _d],
    "prg_cpd_set_resid_sweep": [_vp, _i],
    "prg_cpd_init_sums": [_vp],
    "prg_cpd_init_params": [_vp, _vp],
    "prg_cpd_estep": [_vp, _d],
    "prg_cpd_estep_timed": [_vp, _d, _vp],
    "prg_cpd_pair_counts": [_vp, _c.POINTER(_d), _c.POINTER(_d)],
    "prg_cpd_mstep": [_vp, _i, _i],
    "prg_cpd_get_params": [_vp, _vp],
    "prg_cpd_set_params": [_vp, _vp],
    "prg_cpd_get_moments": [_vp, _vp],
    "prg_cpd_get_estep": [_vp, _vp, _vp, _vp],
    "prg_cpd_get_tsource": [_vp, _vp],
    "prg_cpd_moments_from_estep": [_vp, _vp, _vp, _vp],
    "prg_cpd_set_tuning": [_vp, _i, _i, _i, _i],
    "prg_cpd_nonrigid_build_g": [_vp, _d],
    "prg_cpd_nonrigid_set_solver": [_vp, _i, _i, _d],
    "prg_cpd_nonrigid_rank": [_vp, _c.POINTER(_i)],
    "prg_cpd_nonrigid_get_g": [_vp, _vp],
    "prg_cpd_nonrigid_set_w": [_vp, _vp],
    "prg_cpd_nonrigid_get_w": [_vp, _vp],
    "prg_cpd_nonrigid_apply": [_vp, _vp],
    "prg_cpd_nonrigid_set_priors": [_vp, _vp, _vp, _d],
    "prg_cpd_rowacc_ptr": [_vp, _pp, _c.POINTER(_i64)],
    "prg_cpd_mstep_nonrigid": [_vp, _d],
    "prg_cpd_set_source_weights": [_vp, _vp, _d],
    "prg_cpd_bcpd_build_g": [_vp, _d],
    "prg_cpd_bcpd_solve": [_vp, _d, _d, _vp, _vp, _vp, _vp],
    "prg_gauss_transform_direct": [_i, _vp, _vp, _i64, _vp, _i64, _i, _vp, _i, _d, _vp],
    "prg_squared_kernel_sum": [_i, _vp, _vp, _i64, _vp, _i64, _i, _c.POINTER(_d)],
    "prg_rbf_kernel": [_i, _vp, _vp, _i64, _vp, _i64, _i, _d, _vp],
    "prg_inverse_multiquadric_kernel": [_i, _vp, _vp, _i64, _vp, _i64, _i, _d, _vp],
    "prg_nn_mean_distance": [_i, _vp, _vp, _i64, _vp, _i64, _i, _c.POINTER(_d)],
    "prg_lattice_set_splat_mode": [_i],
    "prg_ph_create": [_pp, _i, _vp],
    "prg_ph_destroy": [_vp],
    "prg_ph_init": [_vp, _vp, _i64, _i, _i],
    "prg_ph_lattice_size": [_vp, _c.POINTER(_i)],
    "prg_ph_filter": [_vp, _vp, _i, _vp],
    "prg_fr_create": [_pp, _i, _vp],
    "prg_fr_destroy": [_vp],
    "prg_fr_set_source": [_vp, _vp, _i64, _i],
    "prg_fr_set_target": [_vp, _vp, _i64, _i],
    "prg_fr_set_state": [_vp, _vp, _vp, _d],
    "prg_fr_estep": [_vp, _d, _c.POINTER(_i), _c.POINTER(_i)],
    "prg_fr_get_estep": [_vp, _vp, _vp, _vp],
    "prg_fr_mstep": [_vp, _d, _i, _d, _vp],
    "prg_fr_get_state": [_vp, _vp],
    "prg_fr_set_target_normals": [_vp, _vp],
    "prg_fr_get_nx": [_vp, _vp],
    "prg_fr_mstep_pt2pl": [_vp, _d, _i, _d, _vp],
    "prg_fr_mstep_from_arrays": [_i, _vp, _vp, _i64, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _vp],
    "prg_kabsch_weighted": [_i, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.argtypes = _args
    _fn.restype = _i
lib.prg_last_error.argtypes = []
lib.prg_last_error.restype = _c.c_char_p


def last_error():
    msg = lib.prg_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status):
    """Map a prg_status to the exception the reference raises in the same situation."""
    if status == PRG_OK:
        return
    msg = last_error()
    if status == PRG_ERR_INVALID:
        raise ValueError(msg)
    if status == PRG_ERR_NOMEM:
        raise MemoryError(msg)
    raise ProbregHipError("libprobreg_hip status %d: %s" % (status, msg))


def device_count():
    n = _i(0)
    st = lib.prg_device_count(_c.byref(n))
    if st != PRG_OK:
        return 0
    return int(n.value)


def require_gpu():
    n = device_count()
    if n <= 0:
        raise ProbregHipError(
            "probreg_amd needs an AMD GPU (gfx950): hipGetDeviceCount reported none (%s). "
            "There is no CPU fallback." % last_error()
        )
    return n


def ptr(a):
    """Raw pointer of a C-contiguous numpy array or of a torch tensor (host or device)."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return _vp(a.data_ptr())
    return _vp(a.ctypes.data)
