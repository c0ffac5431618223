#!/usr/bin/env python
"""What the per-iteration collective costs a rank, both ways, on ONE rank (no peers: launch + kernel + stream handling, not the
xGMI latency): 2000 back-to-back SUM all-reduces of the 32-double moment block
  (a) issued by the library (prg_comm_all_reduce_f64: ncclAllReduce on the plan's stream, one ctypes call),
  (b) through torch.distributed (backend nccl = RCCL) on torch's current stream,
  (c) through torch.distributed under an ExternalStream switch (what dist.all_reduce_sum_ does when the plan's stream is not current),
and the C1 registration (50 iterations, tol < 0) with (a) inside prg_cpd_iterate, with (b) from the Python loop, and with no collective."""
import os
import sys
import time

os.environ["PROBREG_FORCE_DIST"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as tdist  # noqa: E402
from probreg_amd import cpd, dist, synthetic  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
comm = dist.native_comm(0)
assert comm is not None
buf = torch.zeros(32, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
N = 2000


def timed(fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6, t_host / N * 1e6


side = torch.cuda.Stream()
ext = torch.cuda.ExternalStream(side.cuda_stream)


def via_external():
    with torch.cuda.stream(ext):
        tdist.all_reduce(buf)


print("32-double SUM all-reduce on a one-rank RCCL communicator, %d calls back to back: us per call (host enqueue time alone)" % N)
print("  library (prg_comm_all_reduce_f64, plan's stream)      %7.2f  (%6.2f)" % timed(lambda: comm.all_reduce_f64_(buf, st)))
print("  torch.distributed.all_reduce, current stream          %7.2f  (%6.2f)" % timed(lambda: tdist.all_reduce(buf)))
print("  torch.distributed.all_reduce under an ExternalStream  %7.2f  (%6.2f)" % timed(via_external))

src, tgt, _ = synthetic.rigid_pair(100000, seed=0)


def registration(label, native, group):
    os.environ["PROBREG_NATIVE_RCCL"] = "1" if native else "0"
    dist.reset_native_comms()
    if not group:  # no process group at all: the plain single-GPU path
        dist.initialized = lambda: False
    reg = cpd.RigidCPD(src)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = reg.registration(tgt, maxiter=50, tol=-1.0)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("  %-58s %8.2f ms for 50 iterations incl. upload (sigma2 %.9e)" % (label, 1e3 * best, res.sigma2))


print("C1 registration, 50 EM iterations, tol < 0 (nothing read back until the end):")
registration("library-side ncclAllReduce inside prg_cpd_iterate", True, True)
registration("torch.distributed all_reduce from the Python loop", False, True)
_init = dist.initialized
registration("no collective (one process, prg_cpd_iterate)", False, False)
dist.initialized = _init
dist.reset_native_comms()
tdist.destroy_process_group()
