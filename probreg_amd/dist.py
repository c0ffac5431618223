"""Target sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The reference is single-process; this is the new data-parallel layer of SURVEY.md section 8(e):
every rank keeps the whole source cloud, owns a contiguous block of target rows and all-reduces
the 32-double moment block once per EM iteration (rigid / affine), or the per-point fp64 block
(non-rigid).  With the ``gloo`` backend the same helpers work on CPU tensors (used by the tests).
"""
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


def all_reduce_sum_(tensor):
    """In-place SUM all-reduce of a torch tensor (device tensor -> RCCL over xGMI, CPU tensor -> gloo)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():  # also with one rank: keeps the collective path exercised
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
