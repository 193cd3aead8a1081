"""Device-resident per-object perception (SURVEY.md 8 rows f-2 / f-3 / a1 together).

The reference's driver (examples/pybullet_gto_planning.py:176-190) builds two ``DepthPointCloud`` objects from one depth
image, sizes the grid from the first one's points and asks each for its cost field at the grid's voxel centres; the two
fields then travel into ``IKSolver.solve_ik`` and ``GTOPlanner.plan_goalset``.  Through the same calls, this module keeps
everything on the GPU: ``DepthPointCloud.points``, ``GTORobotModel.workspace_points`` and ``get_sdf_cost(...)`` hand out
lazy stand-ins; the first consumer that needs a scene makes ONE ``gto_scene_from_depth`` call (image up once, both
fields, voxel records and distance fields resident) and every solver handle shares that scene.  A stand-in turns into the
numpy array the reference would have returned the moment anything treats it as one (``np.asarray``, arithmetic,
indexing), so code outside this package sees no difference but the time.
"""
from __future__ import annotations

import numpy as np

DEPTH_SCENE = 2  # scene id of the resident depth scene on the robot model's utility handle (0: a solver's own, 1: scratch)


class _LazyArray:
    """Base of the stand-ins: behaves as the array it stands for once anybody looks."""
    _value = None

    def _materialize(self):
        raise NotImplementedError

    def __array__(self, dtype=None, copy=None):
        if self._value is None:
            self._value = self._materialize()
        return self._value if dtype is None else self._value.astype(dtype, copy=False)

    def __getattr__(self, name):  # shape, dtype, min, reshape, ...: whatever the real array has
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.__array__(), name)

    def __getitem__(self, k):
        return self.__array__()[k]

    def __len__(self):
        return len(self.__array__())


class LazyCloudPoints(_LazyArray):
    """``DepthPointCloud.points``: world points of the valid pixels (N, 3)."""

    def __init__(self, dpc):
        self.dpc = dpc

    def _materialize(self):
        return self.dpc._points_now()


class LazyWorkspacePoints(_LazyArray):
    """``GTORobotModel.workspace_points`` of a grid that is sized from a depth cloud which is still on the device."""

    def __init__(self, robot):
        self.robot = robot

    def _materialize(self):
        self.robot._resolve_depth_field()
        return self.robot._workspace_points_now()


class LazyCostField(_LazyArray):
    """``DepthPointCloud.get_sdf_cost(robot.workspace_points)``: a float32 cost per voxel, resident in a scene."""

    def __init__(self, dpc, robot, epsilon, w_inside):
        self.dpc, self.robot, self.epsilon, self.w_inside = dpc, robot, float(epsilon), float(w_inside)

    def ensure_scene(self):
        """(handle, scene id) of the resident scene that holds this field: built on first use from this cloud's image and
        mask (the cloud of all pixels is the same image without the mask, so one build serves both fields)."""
        return self.robot._depth_scene_for(self.dpc, self.epsilon, self.w_inside)

    def _materialize(self):
        h, sid = self.ensure_scene()
        c_all, c_obs = h.scene_fields(sid)
        return c_all if self.dpc.target_mask is None else c_obs


def same_image(a, b) -> bool:
    return a is b or (a.shape == b.shape and np.array_equal(a, b))
