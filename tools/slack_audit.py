#!/usr/bin/env python
"""Actual GPU-vs-reference errors of the parity tests that grant more than the north-star tolerance (audit tool)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN_DIR, Golden, rel_err  # noqa: E402
from probreg_amd import cpd, filterreg  # noqa: E402

fr = Golden(os.path.join(GOLDEN_DIR, "filterreg_golden.npz"))
for name in fr.group("pt2pl"):
    c = fr.case("pt2pl/" + name)
    kw = {k[4:]: c[k] for k in c if k.startswith("arg_")}
    if "maxiter" in kw:
        kw["maxiter"] = int(kw["maxiter"])
    if "update_sigma2" in kw:
        kw["update_sigma2"] = bool(kw["update_sigma2"])
    res = filterreg.registration_filterreg(c["source"], c["target"], target_normals=c["normals"], objective_type="pt2pl", **kw)
    print("pt2pl %-28s rot %.2e t %.2e sigma2 %.2e q %.2e" % (
        name, rel_err(res.transformation.rot, c["out_rot"]), np.max(np.abs(res.transformation.t - c["out_t"])),
        abs(res.sigma2 - c["out_sigma2"]) / c["out_sigma2"], abs(res.q - c["out_q"]) / abs(c["out_q"])))
ms = Golden(os.path.join(GOLDEN_DIR, "mstep_golden.npz"))
c = ms.case("filterreg/synth_pt2pl_update")
from probreg_amd import transformation as tf  # noqa: E402
es = filterreg.EstepResult(c["m0"], c["m1"], c.get("m2"), c.get("nx"))
r = filterreg.RigidFilterReg._maximization_step(c["t_source"], c["target"], es, tf.RigidTransformation(c["rot_p"], c["t_p"]),
                                                c["sigma2"], c["w"], objective_type="pt2pl")
print("pt2pl single M-step: rot %.2e t %.2e sigma2 %.2e q %.2e" % (
    rel_err(r.transformation.rot, c["out_rot"]), np.max(np.abs(r.transformation.t - c["out_t"])),
    abs(r.sigma2 - c["out_sigma2"]) / c["out_sigma2"], abs(r.q - c["out_q"]) / abs(c["out_q"])))
cg = Golden(os.path.join(GOLDEN_DIR, "cpd_golden.npz"))
for name in ("bunny_nonrigid_k5", "synth_nonrigid_1k_k5", "fish_nonrigid_default"):
    c = cg.case("reg/" + name)
    kw = {k[4:]: c[k] for k in c if k.startswith("arg_")}
    if "maxiter" in kw:
        kw["maxiter"] = int(kw["maxiter"])
    res = cpd.registration_cpd(c["source"], c["target"], "nonrigid", **kw)
    ts = res.transformation.transform(c["source"])
    ext = np.max(np.abs(c["out_tsource"] - c["out_tsource"].mean(0)))
    print("nonrigid %-24s T %.2e sigma2 %.2e" % (name, np.max(np.abs(ts - c["out_tsource"])) / ext,
                                               abs(res.sigma2 - c["out_sigma2"]) / c["out_sigma2"]))
gc = Golden(os.path.join(GOLDEN_DIR, "cpd_constrained_golden.npz"))
for name in gc.group("reg"):
    c = gc.case("reg/" + name)
    kw = {}
    if "arg_maxiter" in c:
        kw["maxiter"] = int(c["arg_maxiter"])
    if "arg_tol" in c:
        kw["tol"] = float(c["arg_tol"])
    res = cpd.registration_cpd(c["source"], c["target"], "nonrigid_constrained", alpha=float(c["alpha"]),
                               idx_source=c["idx_source"], idx_target=c["idx_target"], **kw)
    ts = res.transformation.transform(c["source"])
    ext = np.max(np.abs(c["out_tsource"] - c["out_tsource"].mean(0)))
    print("constrained %-24s alpha %.0e T %.2e sigma2 %.2e" % (name, float(c["alpha"]), np.max(np.abs(ts - c["out_tsource"])) / ext,
                                                            abs(res.sigma2 - c["out_sigma2"]) / c["out_sigma2"]))
