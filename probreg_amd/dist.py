"""Target sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The reference is single-process; this is the new data-parallel layer of SURVEY.md section 8(e):
every rank keeps the whole source cloud, owns a spatially compact block of target rows (one cell of a
recursive bisection of the target) and all-reduces
the 32-double moment block once per EM iteration (rigid / affine), or the per-point fp64 block
(non-rigid).  With the ``gloo`` backend the same helpers work on CPU tensors (used by the tests).

The per-iteration collective itself is issued by the library (``NativeComm``: RCCL bound inside libprobreg_hip.so, on the
plan's stream - an EM iteration is enqueue-only) whenever the process group's backend is ``nccl``; ``torch.distributed``
then only carries the one-off hand-over of the communicator id.  The torch path stays for ``gloo`` (CPU tests, the
two-ranks-on-one-GPU rig) and as the fallback when RCCL cannot be bound.
"""
import os
import warnings
import weakref

import numpy as np


def world():
    """(rank, world_size) of the calling process; (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:  # pragma: no cover
        pass
    return 0, 1


def initialized():
    """True when torch.distributed has a process group (even a single-rank one)."""
    try:
        import torch.distributed as dist

        return bool(dist.is_available() and dist.is_initialized())
    except ImportError:  # pragma: no cover
        return False


def shard_bounds(n, rank, world_size):
    """Contiguous near-equal split of ``n`` rows: rows [lo, hi) belong to ``rank``."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of size %d" % (rank, world_size))
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def _spread21(v):
    """21 bits -> every third bit (numpy uint64), the interleave of a 3-D Morton code."""
    v = v & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def morton_order(points):
    """Indices that sort ``points`` (n, 2|3) along the Z-curve of one isotropic 21-bit grid (deterministic)."""
    p = np.asarray(points, dtype=np.float64)
    lo = p.min(axis=0)
    ext = float((p.max(axis=0) - lo).max())
    scale = 2097151.0 / ext if ext > 0.0 else 0.0
    q = ((p - lo) * scale).astype(np.uint64)
    code = np.zeros(p.shape[0], dtype=np.uint64)
    for k in range(p.shape[1]):
        code |= _spread21(q[:, k]) << np.uint64(k)
    return np.argsort(code, kind="stable")


def _shard_start(n, rank, world_size):
    base, rem = divmod(int(n), int(world_size))
    return rank * base + min(rank, rem)


def bisection_shards(points, world_size):
    """Row indices of every rank's shard: recursive bisection of the cloud across the widest axis of the current cell, the
    ranks split in halves and the points in proportion (shard sizes are ``shard_bounds``'s: near-equal counts).  Every shard is
    ONE axis-aligned cell of the cloud for any world size - a contiguous run of a space-filling curve can be two pieces
    either side of one of the curve's jumps.  Deterministic (a function of the cloud alone)."""
    p = np.asarray(points, dtype=np.float64)
    n = p.shape[0]
    out = [None] * int(world_size)
    stack = [(np.arange(n), 0, int(world_size))]
    while stack:
        idx, r0, r1 = stack.pop()
        if r1 - r0 == 1:
            out[r0] = np.sort(idx)
            continue
        mid = (r0 + r1) // 2
        cnt = _shard_start(n, mid, world_size) - _shard_start(n, r0, world_size)
        q = p[idx]
        ax = int(np.argmax(q.max(axis=0) - q.min(axis=0))) if len(idx) else 0
        # (a selection, not a sort: O(n) per cell; numpy's introselect is a function of its input alone, so every rank cuts alike)
        order = np.argpartition(q[:, ax], cnt) if 0 < cnt < len(idx) else np.arange(len(idx))
        stack.append((idx[order[:cnt]], r0, mid))
        stack.append((idx[order[cnt:]], mid, r1))
    return out


def spatial_shard(target, rank, world_size):
    """Row indices of ``target`` owned by ``rank``: one cell of a recursive bisection of the cloud (``bisection_shards``).

    Every rank holds the whole target (the reference API passes full arrays), so all ranks derive the same cut
    and take their part of it.  A shard is then a spatially compact patch: the culled sweeps of a rank skip
    everything far from its patch, exactly as a single GPU does for one wave's points - a shard in the caller's
    order would be spread over the whole object and lose most of the culling in late EM iterations.  With one
    rank the order is left alone (the plan sorts on upload anyway).  ``PROBREG_SHARD_CUT=morton`` restores the cut of
    rounds 2 - 5 (contiguous runs of the target's Z-curve) for A/B runs.
    """
    n = np.asarray(target).shape[0]
    lo, hi = shard_bounds(n, rank, world_size)
    if world_size == 1:
        return np.arange(lo, hi)
    if os.environ.get("PROBREG_SHARD_CUT", "") == "morton":
        return morton_order(target)[lo:hi]
    return bisection_shards(target, world_size)[rank]


def all_reduce_sum_(tensor, stream=None):
    """In-place SUM all-reduce of a torch tensor (device tensor -> RCCL over xGMI, CPU tensor -> gloo).

    ``stream``: the raw hipStream_t (int) the tensor's producer / consumer kernels run on.  torch.distributed orders
    a collective against torch's CURRENT stream, so when the plan was created under a different stream than the one
    current now, the collective is issued under that stream instead - otherwise it would race the plan's kernels.
    """
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():  # also with one rank: keeps the collective path exercised
        if stream is not None and getattr(tensor, "is_cuda", False):
            import torch

            if int(torch.cuda.current_stream(tensor.device).cuda_stream) != int(stream):
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=tensor.device)):
                    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
                return tensor
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def all_reduce_sum_numpy(arr):
    """SUM all-reduce of a (small) float64 numpy array through a CPU / device staging tensor."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


# ----------------------------------------------------------------------------------------------------------------
# the library's own RCCL communicator (include/probreg_hip.h: prg_comm_*)
# ----------------------------------------------------------------------------------------------------------------
class NativeComm(object):
    """An RCCL communicator owned by libprobreg_hip.so (``prg_comm_create``): what ``CpdPlan.set_comm`` takes."""

    def __init__(self, id_bytes, rank, nranks, device):
        import ctypes

        from . import _lib

        self._lib = _lib
        self._h = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * _lib.PRG_COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        _lib.check(_lib.lib.prg_comm_create(ctypes.byref(self._h), buf, int(rank), int(nranks), int(device)))
        self.rank, self.nranks, self.device = int(rank), int(nranks), int(device)
        self._plans = weakref.WeakSet()  # plans whose C side points at this communicator (CpdPlan.set_comm)

    def _attached(self, plan):
        self._plans.add(plan)

    def _detached(self, plan):
        self._plans.discard(plan)

    @staticmethod
    def unique_id():
        import ctypes

        from . import _lib

        buf = (ctypes.c_ubyte * _lib.PRG_COMM_ID_BYTES)()
        _lib.check(_lib.lib.prg_comm_unique_id(buf))
        return bytes(buf)

    def all_reduce_f64_(self, tensor, stream=0):
        """In-place SUM all-reduce of a contiguous fp64 device tensor on the raw hipStream_t ``stream``."""
        import ctypes

        self._lib.check(self._lib.lib.prg_comm_all_reduce_f64(self._h, ctypes.c_void_p(tensor.data_ptr()), int(tensor.numel()),
                                                              ctypes.c_void_p(int(stream))))
        return tensor

    def calls(self):
        import ctypes

        n = ctypes.c_int64(0)
        self._lib.check(self._lib.lib.prg_comm_info(self._h, None, None, ctypes.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # no plan may keep a pointer to a destroyed communicator: its next prg_cpd_estep / prg_cpd_init_sums would
            # all-reduce through freed memory (and the Python side would skip the torch.distributed all-reduce)
            for plan in list(getattr(self, "_plans", ())):
                try:
                    if getattr(plan, "_comm", None) is self and getattr(plan, "_h", None):
                        plan.set_comm(None)
                except Exception:  # pragma: no cover  (a plan that is being torn down itself)
                    pass
            self._lib.lib.prg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


_native = {}  # (device, process-group identity) -> NativeComm or None (None: tried and given up - every rank took the same decision)
_pg_seen = []  # [(group object, generation)]: the object is KEPT so that its id() cannot be handed to a later group
_pg_generation = [0]


def _pg_key():
    """Identity of the default process group (None without one): the collective decision below is taken once per group.

    A generation number, bumped whenever ``torch.distributed.group.WORLD`` is an object this module has not seen: an ``id()``
    alone can be recycled after ``destroy_process_group`` + ``init_process_group``, and the cache would then hand out a
    communicator bound to the dead group.  Entries of earlier groups are closed and dropped as soon as a new one shows up.
    """
    if not initialized():
        return None
    import torch.distributed as tdist

    grp = tdist.group.WORLD
    if not _pg_seen or _pg_seen[-1][0] is not grp:
        _pg_generation[0] += 1
        del _pg_seen[:]
        _pg_seen.append((grp, _pg_generation[0]))
        for key in [k for k in _native if k[1] is not None]:  # communicators of a group that is gone
            c = _native.pop(key)
            if c is not None:
                c.close()
    return (_pg_seen[-1][1], tdist.get_backend(), tdist.get_world_size())


def native_comm(device):
    """The library-side communicator of this process for ``device``, created on first use - or None when the collective
    stays with torch.distributed.

    PROBREG_NATIVE_RCCL: unset - native whenever a process group with the ``nccl`` backend exists; ``0`` - never;
    ``1`` - also for a single process without any process group (a one-rank communicator: tests, measurements).
    Creation is collective: rank 0 draws the id, ``broadcast_object_list`` hands it round, every rank joins, a test
    all-reduce of (1, rank + 1) must come back as (N, N (N + 1) / 2) on every rank - the verdicts are all-reduced (MIN)
    so that all ranks use the same path.
    """
    device = int(device)
    key = (device, _pg_key())
    if key in _native:
        return _native[key]
    mode = os.environ.get("PROBREG_NATIVE_RCCL", "")
    if key[1] is None and mode != "1":
        # no process group (yet) and no request for a one-rank communicator: nothing to decide, and nothing to remember - a
        # group created later is a new key, and EVERY rank then enters the collective set-up below together (a rank that had
        # cached "None" from a single-process registration before init_process_group would issue other collectives than its peers)
        return None
    comm = None
    if mode != "0":
        import torch

        rank, nranks = world()
        have_pg = initialized()
        if have_pg:
            import torch.distributed as tdist

            wanted = tdist.get_backend() == "nccl"
        else:
            wanted = mode == "1"
        if wanted:
            import ctypes

            from . import _lib

            def agree(flag):  # every rank takes the same branch: the verdicts are min-reduced over the process group
                if not have_pg:
                    return flag
                v = torch.tensor([flag], dtype=torch.int32, device="cuda:%d" % device)
                tdist.all_reduce(v, op=tdist.ReduceOp.MIN)
                return int(v.item())

            why = ""
            # 1. can every rank bind librccl at all?  (a rank that cannot must not leave the others waiting in a collective)
            ver = ctypes.c_int(0)
            ok = 1 if _lib.lib.prg_comm_available(ctypes.byref(ver)) == _lib.PRG_OK else 0
            if not ok:
                why = _lib.last_error()
            elif ver.value and not (20000 <= ver.value < 30000):
                # comm.hip restates the ABI of RCCL / NCCL 2.x (ncclUniqueId by value, ncclFloat64 = 8, ncclSum = 0)
                ok, why = 0, "librccl reports version code %d, the restated ABI is that of 2.x" % ver.value
            ok_all = agree(ok)
            ident = None
            if ok_all:
                # 2. rank 0 draws the id; EVERY rank takes part in the broadcast (None tells the others that it failed)
                if rank == 0:
                    try:
                        ident = NativeComm.unique_id()
                    except Exception as e:
                        why = "%s: %s" % (type(e).__name__, e)
                if have_pg:
                    box = [ident]
                    tdist.broadcast_object_list(box, src=0)
                    ident = box[0]
                ok_all = 1 if ident is not None else 0
            if ok_all:
                # 3a. what could make ONE rank raise before it enters ncclCommInitRank (the others would wait in it for good):
                # checked locally and agreed on first
                pre = 1 if (len(ident) == _lib.PRG_COMM_ID_BYTES and 0 <= device < torch.cuda.device_count()) else 0
                if not pre:
                    why = "bad communicator id (%d bytes) or device %d" % (len(ident), device)
                ok_all = agree(pre)
            if ok_all:
                # 3b. join (collective), then the test all-reduce and a last agreement
                ok = 1
                try:
                    comm = NativeComm(ident, rank, nranks, device)
                    with torch.cuda.device(device):
                        probe = torch.tensor([1.0, rank + 1.0], dtype=torch.float64, device="cuda:%d" % device)
                        comm.all_reduce_f64_(probe, torch.cuda.current_stream().cuda_stream)
                        got = probe.cpu().tolist()
                    if got != [float(nranks), nranks * (nranks + 1) / 2.0]:
                        ok, why = 0, "test all-reduce returned %r" % (got,)
                except Exception as e:
                    ok, why = 0, "%s: %s" % (type(e).__name__, e)
                ok_all = agree(ok)
            if not ok_all:
                if why:
                    warnings.warn("probreg_amd: library-side RCCL communicator unavailable (%s); the per-iteration "
                                  "all-reduce goes through torch.distributed" % why)
                if comm is not None:
                    comm.close()
                comm = None
    _native[key] = comm
    return comm


def reset_native_comms():
    """Drop the cached communicators (before ``destroy_process_group`` / in tests)."""
    for c in _native.values():
        if c is not None:
            c.close()
    _native.clear()
    del _pg_seen[:]


import atexit  # noqa: E402

# communicators are torn down while the HIP / RCCL runtimes are still up, not by garbage collection at interpreter shutdown
atexit.register(reset_native_comms)
