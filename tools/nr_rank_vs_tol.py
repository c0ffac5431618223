import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from probreg_amd import cpd, synthetic
from probreg_amd.engine import CpdPlan
n = 50000
src, tgt = synthetic.nonrigid_pair(n, seed=0)
ref = None
for tol in (1e-14, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8):
    plan = CpdPlan()
    plan.set_options(sort_source=True, sort_target=True, cull=True)
    plan.set_source(src)
    plan.set_nonrigid_solver(1, 0, tol)
    t0 = time.perf_counter(); plan.build_g(2.0); torch.cuda.synchronize(); tb = time.perf_counter() - t0
    r = plan.nonrigid_rank()
    plan.set_target(tgt, n_global=n); plan.init_sums(); plan.init_params(None); plan.set_w(np.zeros_like(src))
    for it in range(6):
        plan.estep(0.0); plan.mstep_nonrigid(2.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(10):
        plan.mstep_nonrigid(2.0)
    torch.cuda.synchronize(); tm = (time.perf_counter() - t0) / 10
    s2 = plan.get_params()[13]; T = plan.nonrigid_apply()
    if ref is None: ref = (s2, T)
    ext = np.max(np.abs(ref[1] - ref[1].mean(0)))
    print("tol %.0e rank %d build %.1f ms mstep %.3f ms sigma2 rel diff %.2e T diff/extent %.2e" % (tol, r, tb*1e3, tm*1e3, abs(s2-ref[0])/ref[0], np.max(np.abs(T-ref[1]))/ext))
    plan.close()
