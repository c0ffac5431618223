"""The C/OpenMP E-step restatement (cpu_baseline of bench.py) against the numpy oracle and the
reference-produced E-step fixtures.  CPU only."""
import numpy as np

from oracle import cpd_c, cpd_numpy as co
from conftest import rel_err


def test_c_estep_matches_reference_fixtures(cpd_golden):
    for name in cpd_golden.group("estep"):
        c = cpd_golden.case("estep/" + name)
        pt1, p1, px, n_p = cpd_c.expectation_step(c["t_source"], c["target"], c["sigma2"], c["w"])
        assert np.max(np.abs(pt1 - c["pt1"])) < 1e-12, name
        assert rel_err(p1, c["p1"]) < 1e-12, name
        assert rel_err(px, c["px"]) < 1e-12, name
        assert abs(n_p - c["n_p"]) < 1e-9, name


def test_c_estep_matches_numpy_oracle_seeded():
    rng = np.random.default_rng(2)
    a, b = rng.normal(size=(700, 3)), rng.normal(size=(900, 3)) * 1.1
    want = co.expectation_step(a, b, 0.07, 0.15)
    pt1, p1, px, n_p = cpd_c.expectation_step(a, b, 0.07, 0.15)
    assert rel_err(pt1, want.pt1) < 1e-12 and rel_err(p1, want.p1) < 1e-12 and rel_err(px, want.px) < 1e-12
    assert cpd_c.threads() >= 1
