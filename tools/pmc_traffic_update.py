#!/usr/bin/env python
"""profiles/<tag>_*_pmc.txt (tools/profile_round.sh) -> profiles/pmc_traffic.json, the file bench.py's `roofline.traffic` reads.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB; FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests are tallied as 64).   usage: pmc_traffic_update.py [tag]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = ([a for a in sys.argv[1:] if not a.startswith("--")] or ["r6"])[0]


def table(name):
    out = {}
    path = os.path.join(ROOT, "profiles", "%s_%s_pmc.txt" % (tag, name))
    if not os.path.exists(path):
        return None
    for line in open(path):
        f = line.split()
        if len(f) >= 4 and f[-3] in ("FETCH_SIZE", "WRITE_SIZE"):  # (kernel names may contain spaces: "k_embed<3, true>")
            out.setdefault(" ".join(f[:-3]), {})[f[-3]] = (float(f[-2]), int(f[-1]))
    return {k: ((2.0 * v.get("FETCH_SIZE", (0, 0))[0] + v.get("WRITE_SIZE", (0, 0))[0]) * 1024.0,
                max(v.get("FETCH_SIZE", (0, 0))[1], v.get("WRITE_SIZE", (0, 0))[1])) for k, v in out.items()}


def sweep_entry(t, what):
    per = {k: int(round(b)) for k, (b, n) in t.items() if k.startswith(("k_rowpass", "k_colpass"))}
    calls = {k: n for k, (b, n) in t.items() if k.startswith(("k_rowpass", "k_colpass"))}

    def mean(prefix):
        ks = [k for k in per if k.startswith(prefix)]
        tot = sum(calls[k] for k in ks)
        return int(round(sum(per[k] * calls[k] for k in ks) / tot)) if tot else None

    def mean_of(keys):
        tot = sum(calls[k] for k in keys)
        return int(round(sum(per[k] * calls[k] for k in keys) / tot)) if tot else None

    fused = [k for k in per if k.startswith("k_colpass_mfma<true")]       # the fused single sweep of a rigid iteration
    resid = [k for k in per if k.startswith(("k_colpass_queue<true", "k_colpass_cull<true", "k_colpass_owner"))]  # ... and the residual-form one (rounds 5, 6)
    dominant = fused + resid + [k for k in per if k.startswith("k_rowpass")]  # what bench.py's roofline is quoted on
    return {"rowpass_hbm_bytes_per_launch": mean("k_rowpass"), "colpass_hbm_bytes_per_launch": mean("k_colpass"),
            "fused_sweep_hbm_bytes_per_launch": mean_of(fused), "resid_sweep_hbm_bytes_per_launch": mean_of(resid),
            "dominant_sweep_hbm_bytes_per_launch": mean_of(dominant),
            "per_kernel_bytes_per_launch": per, "launches_in_profile": calls, "how": what, "round": int(tag[1:])}


how = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh, profiles/%s_%%s_pmc.txt: "
       "averages over every launch of the command); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE doubled per "
       "MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B); row / column pass = launch-weighted mean over the "
       "kernels that ran (matrix-core kernels while sigma2 is large, culled vector kernels afterwards)" % tag)
out = {}
t = table("bench_rigid100k")
if t:
    out["rigid_100k"] = sweep_entry(t, how % "bench_rigid100k")
t = table("affine_200k")
if t:
    out["affine_200k"] = sweep_entry(t, how % "affine_200k")
t = table("nonrigid_50k")
if t:
    out["nonrigid_50k"] = sweep_entry(t, how % "nonrigid_50k")
t = table("filterreg_500k")
if t:
    # one launch of the M-step's terms kernel per EM iteration (round 3 folded k_fr_finish into it; round 4 split it out again)
    iters = max([n for k, (b, n) in t.items() if k.startswith("k_fr_terms")] + [0])  # (k_fr_terms<true>, <false>, _pt2pl)
    total = sum(b * n for k, (b, n) in t.items() if not k.startswith(("k_sks", "k_sums", "k_fr_values")))
    out["filterreg_500k"] = {"iteration_hbm_bytes": int(round(total / iters)) if iters else None, "iterations_in_profile": iters,
                             "how": "sum over every kernel of an EM iteration of (2*FETCH_SIZE + WRITE_SIZE)*1024 x launches, "
                                    "divided by the number of iterations (launches of k_fr_terms); " + how % "filterreg_500k",
                             "round": int(tag[1:])}
# The code the passes were taken at.  [r6] The file is only written for a CLEAN tree of kernel sources: run the passes at a commit
# (tools/profile_round.sh), then this script before anything under probreg_amd/ or include/ changes - with local modifications
# there the traffic would be attributed to a commit that did not produce it, so the script refuses (--allow-dirty overrides and
# records the fact).
import subprocess

head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], universal_newlines=True).strip()
dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "probreg_amd", "include"], universal_newlines=True).strip()
if dirty and "--allow-dirty" not in sys.argv:
    sys.exit("pmc_traffic_update: kernel sources differ from HEAD (%s):\n%s\ncommit (or stash) first, or pass --allow-dirty" % (head, dirty))
out["_taken_at"] = {"commit": head, "kernel_sources_modified_since": bool(dirty), "profiles_tag": tag}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "how"} for k, v in out.items()}, indent=1))
