"""The numpy restatement of the direct Gauss transform / L2 distance (oracle/gauss_numpy.py) against the reference's own
outputs (tests/golden/gauss_golden.npz: probreg/gauss_transform.py and probreg/cost_functions.py run unmodified through
oracle/ref_import.load_gauss)."""
import os

import numpy as np

from conftest import GOLDEN_DIR, Golden
from oracle import gauss_numpy as go

G = Golden(os.path.join(GOLDEN_DIR, "gauss_golden.npz"))


def test_gauss_transform_oracle_matches_reference_outputs():
    names = G.group("gt")
    assert len(names) >= 7
    for name in names:
        c = G.case("gt/" + name)
        got = go.compute(c["source"], c["h"], c["target"], c.get("weights"))
        want = c["out"]
        assert got.shape == want.shape, name
        assert np.max(np.abs(got - want)) <= 1e-12 * max(np.max(np.abs(want)), 1e-300), name


def test_l2_dist_oracle_matches_reference_outputs():
    names = G.group("l2")
    assert len(names) >= 5
    for name in names:
        c = G.case("l2/" + name)
        f, g = go.compute_l2_dist(c["mu_source"], c["phi_source"], c["mu_target"], c["phi_target"], c["sigma"])
        assert abs(f - c["out_f"]) <= 1e-12 * abs(c["out_f"]), name
        assert np.max(np.abs(g - c["out_g"])) <= 1e-12 * np.max(np.abs(c["out_g"])), name
