"""The bench contract on the committed line of the round (the newest profiles/r<N>_bench_default_line_1gpu*.json = an
unprofiled `python bench.py` run on one MI355X): every key the driver and the judge read is there, with the promised meaning."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    paths = glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default_line_1gpu*.json"))
    paths = [p for p in paths if "profiled" not in os.path.basename(p)]
    newest = max(paths, key=lambda p: (int(re.match(r"r(\d+)_", os.path.basename(p)).group(1)), os.path.basename(p)))
    with open(newest) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_contract_keys_and_types():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]      # value = iterations / second of the timed window
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and 0.0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_other_workloads_ride_in_the_same_line():
    d = _line()
    other = d["other_workloads"]
    assert set(other) == {"affine_200k", "nonrigid_50k", "filterreg_500k"}
    for name, w in other.items():
        assert "error" not in w, (name, w.get("error"))
        assert w["value"] > 0 and 0.0 < w["roofline"]["frac"] <= 1.0
    nr = other["nonrigid_50k"]
    assert nr["kernel_factor"]["rank"] > 0                       # the product path is the kernel factor ...
    assert nr["dense_solver"]["max_dT_over_extent"] < 1e-4       # ... and it agrees with the dense fallback
    assert d["parity"]["ok"] is True
    if "parity" in other["affine_200k"]:  # (round 3 on: every workload carries its own GPU-vs-oracle block)
        for name, w in other.items():
            assert w["parity"]["ok"] is True, (name, w["parity"])


def test_roofline_flop_accounting_is_self_consistent():
    """`frac` charges every launch the flop its kernel spends per pair: 21 for the row pass, 19 where the matrix-core row
    pass ran without its residual sums (prg_cpd_last_estep_lean); the window's average and the shares are in the line."""
    r = _line()["roofline"]
    f = r["flop_per_pair"]["isa_count"]
    if "row_lean" not in f:  # (lines of earlier rounds)
        return
    assert f["row"] == 21.0 and f["row_lean"] == 19.0 and f["col"] == 14.0
    assert f["row_lean"] <= f["row_window_average"] <= max(f["row"], f.get("fused", 0.0))  # (round 4: fused single sweep, 22)
    share = r["matrix_core_share"]
    assert 0.0 <= share["row_pass_lean_iterations"] <= share["row_pass_iterations"] <= 1.0
    # achieved = (pairs x flop/pair) / time, per launch: the line's own per-launch figures reproduce it
    pairs, ms = r["pairs_evaluated_per_launch"], r["avg_launch_ms"]
    achieved = pairs * f["row_window_average"] / (ms * 1e-3) / 1e12
    assert abs(achieved - r["achieved"]) < 1e-6 * r["achieved"]
    assert r["frac_with_survey_flops"] < r["frac"] * 20.0 / f["row_lean"] + 1e-12
