"""DepthPointCloud — drop-in for the reference's mesh_to_sdf/depth_point_cloud.py.

The reference back-projects a depth image, builds a scikit-learn KD-tree over the points and queries it
at every voxel centre to produce ``sdf_cost_all`` / ``sdf_cost_obstacle`` (seconds per scene,
examples/pybullet_gto_planning.py:180-190).  Here the back-projection, the exact nearest-neighbour
distance (FP64 on the MI355X: a bounding-box hierarchy over tiles of the depth image walked by packets of 64
neighbouring queries; same bits as an exhaustive search), the visibility test ``is_outside`` and the cost map
run in ``gto_depth_sdf_cost``; the values are bit-identical to the reference (tests/golden/depth_cost.npz).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _capi
from .depth_scene import LazyCloudPoints, LazyCostField, LazyWorkspacePoints


class DepthPointCloud:
    def __init__(self, depth, intrinsic_matrix, camera_pose, target_mask=None, threshold=1.5, device=0):
        self.depth = np.ascontiguousarray(depth, dtype=np.float32)
        self.intrinsic_matrix = np.ascontiguousarray(intrinsic_matrix, dtype=np.float64).reshape(3, 3)
        self.camera_pose = np.ascontiguousarray(camera_pose, dtype=np.float64).reshape(4, 4)
        self.target_mask = None if target_mask is None else np.ascontiguousarray(target_mask, dtype=np.uint8)
        self.height, self.width = self.depth.shape
        self.threshold = float(threshold)
        self.device = device
        # the reference inverts with np.linalg.inv (:34, :128); the same inverses go to the GPU
        self._Kinv = np.ascontiguousarray(np.linalg.inv(self.intrinsic_matrix))
        self._cam_inv = np.ascontiguousarray(np.linalg.inv(self.camera_pose))
        self._lib = _capi.load_library()
        self._points = None

    # ------------------------------------------------------------------ the one GPU call
    def _run(self, query, epsilon=0.02, w_inside=1.0, want_points=False):
        query = np.ascontiguousarray(query, dtype=np.float64).reshape(-1, 3)
        nq = query.shape[0]
        N = self.height * self.width
        sdf = np.empty(nq, dtype=np.float32)
        cost = np.empty(nq, dtype=np.float32)
        inside = np.empty(nq, dtype=np.uint8)
        pts = np.empty((N, 3)) if want_points else None
        valid = np.empty(N, dtype=np.uint8) if want_points else None
        pu8, pf, pd = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)
        p = lambda a, t: None if a is None else a.ctypes.data_as(t)
        rc = self._lib.gto_depth_sdf_cost(self.device, p(self.depth, pf), self.height, self.width, p(self.intrinsic_matrix, pd),
                                          p(self._Kinv, pd), p(self.camera_pose, pd), p(self._cam_inv, pd), p(self.target_mask, pu8),
                                          self.threshold, p(query, pd), nq, float(epsilon), float(w_inside), p(sdf, pf),
                                          p(inside, pu8), p(cost, pf), p(pts, pd), p(valid, pu8))
        if rc != 0:
            raise _capi.GTOError(f"gto_depth_sdf_cost failed ({rc}): {self._lib.gto_last_error(None).decode()}")
        if want_points:
            self._points = pts[valid.astype(bool)]
        return sdf, inside.astype(bool), cost

    # ------------------------------------------------------------------ reference surface
    @property
    def points(self):
        """World points of the valid pixels in pixel order (:21-23).  A lazy stand-in: the array is only brought to the host
        if somebody looks at it (GTORobotModel.setup_points_field takes the bounding box from the device instead)."""
        if self._points is not None:
            return self._points
        return LazyCloudPoints(self)

    def _points_now(self):
        if self._points is None:
            self._run(np.zeros((0, 3)), want_points=True)
        return self._points

    def get_random_surface_points(self, count):
        pts = self._points_now()
        indices = np.random.choice(pts.shape[0], count)
        return pts[indices, :]

    def get_sdf(self, query_points):
        """:56-61 — float32 signed distances."""
        return self._run(query_points)[0]

    def get_sdf_cost(self, query_points, epsilon=0.02, w_inside=1, vis=False):
        """:64-91 — float32 costs (``vis`` is ignored: no viewer here).  Asked at the voxel centres of a grid that was sized
        from a depth cloud (``robot.workspace_points`` still lazy), the answer stays on the device (depth_scene.py)."""
        if isinstance(query_points, LazyWorkspacePoints):
            return LazyCostField(self, query_points.robot, epsilon, w_inside, grid_dpc=query_points.grid_dpc)
        return self._run(query_points, epsilon, w_inside)[2]

    def get_sdf_in_batches(self, query_points, batch_size=1000000):
        """:94-104."""
        query_points = np.asarray(query_points)
        if query_points.shape[0] <= batch_size:
            return self.get_sdf(query_points)
        n_batches = int(math.ceil(query_points.shape[0] / batch_size))
        return np.concatenate([self.get_sdf(pts) for pts in np.array_split(query_points, n_batches)])

    def is_outside(self, points):
        """:126-141."""
        return ~self._run(points)[1]
