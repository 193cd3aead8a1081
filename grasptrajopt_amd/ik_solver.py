"""IKSolver — drop-in for the reference's gto/ik_solver.py, solved on the GPU.

The reference builds a T = 1 OpTaS problem (gto/ik_solver.py:30-76): gripper point matching against
``RT @ gripper_tf`` + ``10 * sum(sdf_cost_obstacle[offsets])`` over every collision link + joint limits,
and hands it to IPOPT (max_iter 50).  Here the same objective goes through ``gto_solve_ik_batch``: one
workgroup per goal pose runs the whole projected Levenberg-Marquardt iteration on the MI355X, so the
candidate grasps of an object are solved in ONE call (``solve_ik_batch``) instead of one IPOPT run each
(examples/pybullet_gto_planning.py:242-272).
"""
from __future__ import annotations

import numpy as np


class IKSolver:
    def __init__(self, robot, link_ee, link_gripper, collision_avoidance=True):
        self.robot = robot
        self.link_ee = link_ee
        self.link_gripper = link_gripper
        self.robot_name = robot.get_name()
        self.gripper_points = robot.surface_pc_map[link_gripper].points
        self.collision_avoidance = collision_avoidance
        self.max_iter = 50  # gto/ik_solver.py:76
        self._handle = None

    def setup_optimization(self):
        """gto/ik_solver.py:30-77: nothing symbolic to build; binds the solver handle."""
        self._handle = self.robot.solver_handle(self.link_ee, self.link_gripper, role="ik")
        self._fe = self.robot.desc.frame_index(self.link_ee)

    # ------------------------------------------------------------------ batched entry point
    def solve_ik_batch(self, q_0, RTs, sdf_cost_obstacle=None, base_position=None):
        """B seeds ``q_0 (B, ndof)`` (or one seed for all) and goal poses ``RTs (B, 4, 4)`` of link_ee.
        Returns (q (B, ndof), err_pos (B,), err_rot_deg (B,), cost (B,), iters (B,), status (B,))."""
        if self._handle is None:
            self.setup_optimization()
        h = self._handle
        RTs = np.asarray(RTs, dtype=np.float64).reshape(-1, 4, 4)
        B = RTs.shape[0]
        q_0 = np.broadcast_to(np.asarray(q_0, dtype=np.float64).reshape(-1, self.robot.ndof), (B, self.robot.ndof))
        base = np.zeros(3) if base_position is None else np.asarray(base_position, dtype=np.float64).reshape(3)
        sid = None
        if self.collision_avoidance:
            if sdf_cost_obstacle is None:
                raise ValueError("collision_avoidance=True needs sdf_cost_obstacle")
            sid = 0  # the IK solver owns its handle
            from .depth_scene import resident_of
            r = resident_of(sdf_cost_obstacle)
            if r is not None:  # resident on the device (depth_scene.py): shared, not uploaded; whichever half of its scene the
                # field is, it is what this solver reads as the obstacle field (shared again on every call: a later build of
                # the resident scene leaves no stale pointer behind)
                h.share_scene(sid, r.handle, r.sid, all_from=r.half, obs_from=r.half)
            else:
                shape, origin, res = self.robot.field_geometry()
                h.set_scene(sid, np.asarray(sdf_cost_obstacle), None, shape, origin, res)
        q, f, iters, status = h.solve_ik_batch(sid, q_0, RTs.reshape(B, 16), base, self.max_iter)
        # errors as the reference reports them (gto/ik_solver.py:88-93)
        tf = h.eval_fk(q)[:, self._fe]
        err_pos = np.linalg.norm(RTs[:, :3, 3] - tf[:, :3, 3], axis=1)
        cosang = (np.einsum("bij,bij->b", RTs[:, :3, :3], tf[:, :3, :3]) - 1.0) / 2.0  # = 2 (q1.q2)^2 - 1
        err_rot = np.degrees(np.arccos(np.clip(cosang, -1.0, 1.0)))
        cost = np.zeros(B)
        if self.collision_avoidance:  # compute_plan_cost of a one-column plan (gto/gto_models.py:204-215)
            _, _, val, _ = h.eval_points(sid, q, base, use_obs=True, want=("val",))
            cost = val.sum(axis=1)
        return q, err_pos, err_rot, cost, iters, status

    # ------------------------------------------------------------------ reference signature
    def solve_ik(self, q_0, RT, sdf_cost_obstacle=None, base_position=None):
        """gto/ik_solver.py:78-110 -> (q (ndof,), err_pos, err_rot_deg, cost)."""
        q, ep, er, c, _, _ = self.solve_ik_batch(np.asarray(q_0, dtype=np.float64).reshape(1, -1), np.asarray(RT)[None],
                                                 sdf_cost_obstacle, base_position)
        return q[0], float(ep[0]), float(er[0]), float(c[0])

    def solve_fk(self, q_0):
        """gto/ik_solver.py:113-114."""
        if self._handle is None:
            self.setup_optimization()
        return self._handle.eval_fk(np.asarray(q_0, dtype=np.float64).reshape(1, -1))[0, self._fe]
