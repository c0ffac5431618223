"""The C-ABI shared library loads on a machine without a GPU and exports exactly what
include/probreg_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "probreg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prg_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from probreg_amd import _lib

    assert os.path.isfile(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(_lib.lib, name), "libprobreg_hip.so does not export %s" % name
    # the Python signature table covers the header (prg_last_error is bound separately)
    missing = [n for n in names if n not in _lib.SIGNATURES and n != "prg_last_error"]
    assert not missing, "no ctypes signature for %s" % missing
    extra = [n for n in _lib.SIGNATURES if n not in names]
    assert not extra, "ctypes signature without a header declaration: %s" % extra


def test_every_declared_symbol_is_mapped_to_the_reference_in_integration_md():
    """INTEGRATION.md section 2 names every entry point with the reference interface it replaces (or says that there is none)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in _declared_symbols() if "`%s`" % n not in text]
    assert not missing, "INTEGRATION.md does not name %s" % missing


def test_version_and_error_channel():
    from probreg_amd import _lib

    assert _lib.lib.prg_version() >= 100
    # invalid arguments are reported through the status + message channel, never a crash
    st = _lib.lib.prg_device_count(None)
    assert st == _lib.PRG_ERR_INVALID
    assert "NULL" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(st)


def test_no_gpu_fails_loudly():
    """On a box without a GPU the product raises instead of silently computing on the CPU."""
    import numpy as np
    from probreg_amd import _lib, cpd

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    x = np.random.default_rng(0).normal(size=(20, 3))
    with pytest.raises(_lib.ProbregHipError):
        cpd.registration_cpd(x, x + 0.1)


def test_product_does_not_import_oracle():
    """Nothing under probreg_amd/ may import or reference the oracle package."""
    pkg = os.path.join(ROOT, "probreg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
                assert "oracle/" not in text or f == "_never_", os.path.join(dirpath, f)
