"""probreg_amd - MI355X-native engine for probreg's CPD / FilterReg EM hot path.

Public modules mirror the reference package (neka-nat/probreg):
``cpd`` (registration_cpd, RigidCPD, AffineCPD, NonRigidCPD), ``filterreg``
(registration_filterreg), ``transformation``, ``math_utils``, ``gauss_transform``,
``gaussian_filtering``, ``cost_functions.compute_l2_dist``.  All arithmetic of the hot path runs in ``csrc/libprobreg_hip.so``.
Sub-modules are imported on first attribute access so that ``import probreg_amd`` itself
never touches the GPU.
"""
import importlib

from .version import __version__

_SUBMODULES = ("cpd", "filterreg", "transformation", "math_utils", "gauss_transform", "gaussian_filtering",
               "cost_functions", "dist", "engine", "synthetic", "bcpd")


def __getattr__(name):
    if name in _SUBMODULES:
        return importlib.import_module("." + name, __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
