"""BasePlanner — drop-in for the reference's gto/base_planner.py, solved on the GPU.

The reference builds an OpTaS problem with T = goal_size (gto/base_planner.py:35-94): a planar base pose
(x, y, theta) shared by all goals, one arm configuration per goal, gripper point matching against
``tf_base @ RT_i @ gripper_tf``, an effort term on the base pose, joint limits and |theta| <= pi, and hands
it to IPOPT (max_iter 100).  Here the same objective goes through ``gto_solve_base_batch``: one workgroup
per goal set runs the whole projected Levenberg-Marquardt iteration on the MI355X, and several goal sets
(the resampling loop of examples/pybullet_gto_planning_mobile.py:185-199) can be solved in ONE call.
"""
from __future__ import annotations

import numpy as np

from .utils import rotZ


class BasePlanner:
    def __init__(self, robot, link_ee, link_gripper):
        self.robot = robot
        self.robot_name = robot.get_name()
        self.link_ee = link_ee
        self.link_gripper = link_gripper
        self.gripper_points = robot.surface_pc_map[link_gripper].points
        self.task_name = "base_pose_estimator"  # gto/base_planner.py:23
        self.max_iter = 100  # gto/base_planner.py:95
        self.goal_size = 1
        self.base_effort_weight = 0.01
        self._handle = None

    def setup_optimization(self, goal_size=1, base_effort_weight=0.01):
        """gto/base_planner.py:35-96: nothing symbolic to build; records the sizes and binds the handle."""
        self.goal_size = int(goal_size)
        self.base_effort_weight = float(base_effort_weight)
        self._handle = self.robot.solver_handle(self.link_ee, self.link_gripper, role="base")
        self._fe = self.robot.desc.frame_index(self.link_ee)
        self._fg = self.robot.desc.frame_index(self.link_gripper)

    # ------------------------------------------------------------------ batched entry point
    def plan_goalset_batch(self, qc, RTs_sets, n_goals=None):
        """B goal sets ``RTs_sets (B, n, 4, 4)`` (ragged through ``n_goals (B,)``) from configuration(s) ``qc``.
        Returns (Q (B, ndof, n), y (B, 3), err_pos (B, n), err_rot_deg (B, n), iters (B,), status (B,))."""
        if self._handle is None:
            self.setup_optimization(np.asarray(RTs_sets).shape[1], self.base_effort_weight)
        h, ndof = self._handle, self.robot.ndof
        RTs_sets = np.asarray(RTs_sets, dtype=np.float64)
        B, n = RTs_sets.shape[:2]
        qc = np.broadcast_to(np.asarray(qc, dtype=np.float64).reshape(-1, ndof), (B, ndof))
        y, q, _, iters, status = h.solve_base_batch(qc, RTs_sets, n_goals, self.base_effort_weight, self.max_iter)
        # errors as the reference reports them (gto/base_planner.py:127-143)
        fr = h.eval_fk(q.reshape(B * n, ndof)).reshape(B, n, -1, 4, 4)
        tf = fr[:, :, self._fg]
        G = np.linalg.inv(fr[:, :, self._fe]) @ tf
        RT_base = np.stack([_base_matrix(v) for v in y])
        RT = RT_base[:, None] @ RTs_sets @ G
        err_pos = np.linalg.norm(RT[..., :3, 3] - tf[..., :3, 3], axis=-1).astype(np.float32)
        cosang = (np.einsum("bnij,bnij->bn", RT[..., :3, :3], tf[..., :3, :3]) - 1.0) / 2.0  # = 2 (q1.q2)^2 - 1
        err_rot = np.degrees(np.arccos(np.clip(cosang, -1.0, 1.0))).astype(np.float32)
        return np.transpose(q, (0, 2, 1)).copy(), y, err_pos, err_rot, iters, status

    def base_collision_cost(self, qc, y):
        """gto/base_planner.py:146-158: robot surface points at qc, seen from the moved base, summed over the
        x-y occupancy grid (robot.setup_occupancy_grid)."""
        if not hasattr(self.robot, "occupancy_grid"):
            raise RuntimeError("call robot.setup_occupancy_grid(points) before planning the base")
        RT_base_inv = np.linalg.inv(_base_matrix(y))
        pts, _ = self.robot.compute_fk_surface_points(np.asarray(qc, dtype=np.float64).reshape(-1), tf_base=RT_base_inv)
        offsets = self.robot.points_to_offsets_occupancy_numpy(pts)
        return float(np.sum(self.robot.occupancy_grid[offsets]))

    # ------------------------------------------------------------------ reference signature
    def plan_goalset(self, qc, RTs):
        """gto/base_planner.py:97-165 -> (Q (ndof, n), y (3,), err_pos (n,), err_rot_deg (n,), collision cost)."""
        RTs = np.asarray(RTs, dtype=np.float64).reshape(-1, 4, 4)
        Q, y, err_pos, err_rot, _, _ = self.plan_goalset_batch(np.asarray(qc, dtype=np.float64).reshape(1, -1), RTs[None])
        cost = self.base_collision_cost(qc, y[0])
        return Q[0], y[0], err_pos[0], err_rot[0], cost


def _base_matrix(y):
    """RT_base of gto/base_planner.py:122-125."""
    M = rotZ(y[2])
    M[0, 3], M[1, 3] = y[0], y[1]
    return M
