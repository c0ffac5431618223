import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


class Golden(object):
    """Nested view of a flat ``a/b/c`` keyed npz."""

    def __init__(self, path):
        self._z = np.load(path)

    def group(self, prefix):
        names = sorted({k[len(prefix) + 1:].split("/")[0] for k in self._z.files if k.startswith(prefix + "/")})
        return names

    def case(self, prefix):
        out = {}
        for k in self._z.files:
            if k.startswith(prefix + "/"):
                v = self._z[k]
                out[k[len(prefix) + 1:]] = v if v.ndim else v.item()
        return out


@pytest.fixture(scope="session")
def cpd_golden():
    return Golden(os.path.join(GOLDEN_DIR, "cpd_golden.npz"))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / den


@pytest.fixture(scope="session")
def mstep_golden():
    """Single M-step calls of the reference on explicit E-step arrays (tests/golden/make_golden.py mstep)."""
    return Golden(os.path.join(GOLDEN_DIR, "mstep_golden.npz"))


@pytest.fixture(scope="session")
def fr_golden():
    return Golden(os.path.join(GOLDEN_DIR, "filterreg_golden.npz"))


@pytest.fixture(scope="session")
def feature_golden():
    """Feature-space lattices (d = 5, 33) and FilterReg runs with a non-identity feature_fn, from the reference
    (tests/golden/make_golden.py features)."""
    return Golden(os.path.join(GOLDEN_DIR, "feature_lattice_golden.npz"))


def golden_feature_map(case):
    """The deterministic position -> feature map stored with a feature-FilterReg fixture (make_golden.feature_map)."""
    a, ph = case["feat_a"], case["feat_phase"]

    def fn(x):
        x = np.asarray(x, dtype=np.float64)
        return np.concatenate([x, 0.3 * np.sin(x @ a + ph)], axis=1)

    return fn
