"""Transformation result types of the registration API (reference probreg/transformation.py:17-102).

These are small host-side value objects (a 3x3 matrix, a vector, an M x D weight matrix); the
per-iteration transform of the *source cloud* inside the EM loop runs fused on the GPU
(``k_transform_linear`` / ``k_gw`` in csrc/), not through these classes.
"""
import abc

import numpy as np


def _is_vector3d(points):
    # open3d is optional: duck-type o3.utility.Vector3dVector (reference transformation.py:23-26)
    return type(points).__name__ == "Vector3dVector"


class Transformation(abc.ABC):
    def __init__(self, xp=np):
        self.xp = xp

    def transform(self, points, array_type=None):
        if _is_vector3d(points) or (array_type is not None and isinstance(points, array_type)):
            return type(points)(self._transform(np.asarray(points)))
        return self._transform(points)

    @abc.abstractmethod
    def _transform(self, points):
        return points


class RigidTransformation(Transformation):
    """x -> scale * rot @ x + t   (reference transformation.py:33-60)."""

    def __init__(self, rot=np.identity(3), t=np.zeros(3), scale=1.0, xp=np):
        super(RigidTransformation, self).__init__(xp)
        self.rot = rot
        self.t = t
        self.scale = scale

    def _transform(self, points):
        return self.scale * np.dot(points, self.rot.T) + self.t

    def inverse(self):
        return RigidTransformation(self.rot.T, -np.dot(self.rot.T, self.t) / self.scale, 1.0 / self.scale)

    def __mul__(self, other):
        return RigidTransformation(
            np.dot(self.rot, other.rot), self.t + self.scale * np.dot(self.rot, other.t), self.scale * other.scale
        )


class AffineTransformation(Transformation):
    """x -> b @ x + t   (reference transformation.py:63-78)."""

    def __init__(self, b=np.identity(3), t=np.zeros(3), xp=np):
        super(AffineTransformation, self).__init__(xp)
        self.b = b
        self.t = t

    def _transform(self, points):
        return np.dot(points, self.b.T) + self.t


class CombinedTransformation(Transformation):
    """x -> scale * rot @ (x + v) + t : non-rigid displacement first, similarity second (reference
    transformation.py:105-121, the result type of BCPD).  ``v`` is one row per point (or 0)."""

    def __init__(self, rot=np.identity(3), t=np.zeros(3), scale=1.0, v=0.0):
        super(CombinedTransformation, self).__init__()
        self.rigid_trans = RigidTransformation(rot, t, scale)
        self.v = v

    def _transform(self, points):
        return self.rigid_trans._transform(points + self.v)


class NonRigidTransformation(Transformation):
    """y_m -> y_m + (G W)_m on the control points it was built with (reference transformation.py:81-102).

    ``g`` is the float32 Gaussian kernel matrix of the control points.  When the object comes out
    of ``NonRigidCPD`` the GPU plan holds the kernel as its factor ``G = F F^T`` (or, when the kernel is too narrow to
    factor, as the matrix itself); ``.g`` evaluates / downloads the M x M float32 matrix on first access (M*M*4 bytes).
    """

    def __init__(self, w, points, beta=2.0, xp=np, _plan=None, _plan_points=None):
        super(NonRigidTransformation, self).__init__(xp)
        self._points = np.asarray(points)
        self._beta = beta
        self._plan = _plan
        # the control points as the plan holds them (NonRigidCPD shifts far-from-origin clouds before the float32 upload)
        self._plan_points = self._points if _plan_points is None else np.asarray(_plan_points)
        self._g = None
        self.w = w

    @property
    def g(self):
        if self._g is None:
            if self._plan is not None:
                self._g = self._plan.get_g()
            else:
                from . import math_utils as mu

                self._g = mu.rbf_kernel(self._points, self._points, self._beta)
        return self._g

    def close(self):
        """Release the GPU plan an explicit-array M-step (``NonRigidCPD._maximization_step``) cached on this object."""
        cache = self.__dict__.pop("_mstep_plan", None)
        if cache is not None:
            cache[0].close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the library may be gone already
            pass

    def _transform(self, points):
        # same contract as the reference: ``points`` must be the control points the kernel was built on
        if self._plan is not None:
            # G W on the GPU (fp64; through the kernel factor, or the float32 G of the dense fallback), never through a host copy of G
            self._plan.set_w(np.asarray(self.w, dtype=np.float64))
            disp = self._plan.nonrigid_apply() - self._plan_points.astype(np.float32).astype(np.float64)
            return np.asarray(points) + disp
        return points + np.dot(self.g, self.w)
