"""GTOPlanner — same constructor and plan()/plan_goalset() signatures and return values as the
reference (gto/gto_planner.py:21-245); the solve runs on the MI355X solver behind the C ABI."""
from __future__ import annotations

import numpy as np

from . import optas_facade as optas
from .synthetic import standoff_pose
from .utils import interpolate_waypoints


class GTOPlanner:
    def __init__(self, robot, link_ee, link_gripper, collision_avoidance=True, standoff_distance=-0.1,
                 standoff_offset=-10):
        self.T = 50          # gto/gto_planner.py:25
        self.Tmax = 10.0     # :26
        self.dt = self.Tmax / (self.T - 1)  # :27-28
        self.standoff_offset = standoff_offset
        self.standoff_distance = standoff_distance
        self.robot = robot
        self.robot_name = robot.get_name()
        self.link_ee = link_ee
        self.link_gripper = link_gripper
        self.gripper_points = robot.surface_pc_map[link_gripper].points
        self.collision_avoidance = collision_avoidance
        self.max_iter = 100  # :141

    def setup_optimization(self, goal_size=1, use_standoff=False, axis_standoff="x"):
        """gto/gto_planner.py:42-142 with structured cost terms instead of CasADi expressions."""
        builder = optas.OptimizationBuilder(T=self.T, robots=[self.robot])
        builder.add_parameter("qc", self.robot.ndof)
        builder.add_parameter("tf_goal", 16, goal_size)
        builder.add_parameter("sdf_cost_all", self.robot.field_size)
        builder.add_parameter("sdf_cost_obstacle", self.robot.field_size)
        builder.add_parameter("base_position", 3)
        builder.initial_configuration(self.robot_name)
        builder.initial_configuration(self.robot_name, time_deriv=1)
        builder.integrate_model_states(self.robot_name, time_deriv=1, dt=self.dt)
        self.pose_standoff = standoff_pose(self.standoff_distance, axis_standoff)
        builder.add_cost_term("cost_pos", optas.GoalSetPointMatching(
            self.link_ee, self.link_gripper, goal_size, bool(use_standoff), self.pose_standoff, self.standoff_offset))
        if self.collision_avoidance:
            builder.add_cost_term("cost_obstacle", optas.ObstacleField(10.0, self.standoff_offset))
        builder.add_cost_term("min_join_vel", optas.JointVelocity(0.01))
        builder.enforce_model_limits(self.robot_name)
        solver_options = {"ipopt": {"max_iter": self.max_iter, "tol": 1e-15}}
        self.solver = optas.CasADiSolver(builder.build()).setup("ipopt", solver_options=solver_options)

    def _seed_from(self, qc, q_solution):
        data = interpolate_waypoints(np.stack([qc, q_solution]), self.T, self.robot.ndof)
        index = np.array(self.robot.parameter_joint_indexes).astype(np.int32)
        data[:, index] = np.array(qc)[index]
        return data.T

    def plan(self, qc, RT, sdf_cost_obstacle, base_position, q_solution=None, use_standoff=True, axis_standoff="x"):
        """gto/gto_planner.py:145-182.  As in the reference, ``sdf_cost_all`` is not passed here and
        therefore defaults to zeros for the waypoints before the standoff."""
        if hasattr(sdf_cost_obstacle, "resident"):  # a field that is resident on the device: its scene (and with it the grid
            sdf_cost_obstacle.resident()          # geometry the problem is sized by) before anything else
        self.setup_optimization(goal_size=1, use_standoff=use_standoff, axis_standoff=axis_standoff)
        qc = np.asarray(qc, dtype=np.float64)
        tf_goal = np.zeros((16, 1))
        tf_goal[:, 0] = np.asarray(RT, dtype=np.float64).flatten()
        if q_solution is None:
            Q0 = np.diag(qc) @ np.ones((self.robot.ndof, self.T))
        else:
            Q0 = self._seed_from(qc, np.asarray(q_solution, dtype=np.float64))
        self.solver.reset_initial_seed({f"{self.robot_name}/q/x": self.robot.extract_optimized_dimensions(Q0)})
        self.solver.reset_parameters({
            "qc": qc, "tf_goal": tf_goal, "sdf_cost_obstacle": sdf_cost_obstacle, "base_position": base_position,
            f"{self.robot_name}/q/p": self.robot.extract_parameter_dimensions(Q0)})
        solution = self.solver.solve()
        return (solution[f"{self.robot_name}/q"].toarray(), solution[f"{self.robot_name}/dq"].toarray(),
                solution["f"].toarray().flatten())

    def plan_goalset(self, qc, RTs, sdf_cost_all, sdf_cost_obstacle, base_position, q_solutions=None,
                     use_standoff=True, axis_standoff="x", interpolate=True):
        """gto/gto_planner.py:185-245."""
        for fld in (sdf_cost_obstacle, sdf_cost_all):  # fields that are resident on the device: their scene (and with it the
            if hasattr(fld, "resident"):               # grid geometry the problem is sized by) before anything else; the
                fld.resident()                         # obstacle field first: its build holds both fields
        RTs = np.asarray(RTs, dtype=np.float64)
        qc = np.asarray(qc, dtype=np.float64)
        n = RTs.shape[0]
        self.setup_optimization(goal_size=n, use_standoff=use_standoff, axis_standoff=axis_standoff)
        tf_goal = np.zeros((16, n))
        for i in range(n):
            tf_goal[:, i] = RTs[i].flatten()
        # parameters first: the parameter-joint rows of every seed are qc's (gto/gto_planner.py:204-205,234), so the
        # dictionary does not depend on the seed, and the scene is uploaded ONCE for seed scoring and solve
        qp = np.tile(qc[np.array(self.robot.parameter_joint_indexes, dtype=np.int64)][:, None], (1, self.T))
        self.solver.reset_parameters({
            "qc": qc, "tf_goal": tf_goal, "sdf_cost_all": sdf_cost_all, "sdf_cost_obstacle": sdf_cost_obstacle,
            "base_position": base_position, f"{self.robot_name}/q/p": qp})
        if q_solutions is None:
            Q0 = np.diag(qc) @ np.ones((self.robot.ndof, self.T))
        else:
            # seed = interpolation towards the IK solution with the lowest obstacle cost, ties broken by
            # joint distance (gto/gto_planner.py:197-215); all candidates are scored in one GPU call against
            # c_obs of the scene the solve uses
            q_solutions = np.asarray(q_solutions, dtype=np.float64)
            plans = np.stack([self._seed_from(qc, q_solutions[:, i]) for i in range(q_solutions.shape[1])])
            sid = self.solver.ensure_scene()
            cost_all, dist_all = self.solver._handle.plan_cost(sid, plans, base_position)
            self.seed_cost_all, self.seed_dist_all = cost_all, dist_all
            ind = np.lexsort((dist_all, cost_all))
            self.seed_index = int(ind[0])
            if interpolate:
                Q0 = plans[ind[0]]
            else:
                Q0 = np.diag(qc) @ np.ones((self.robot.ndof, self.T))
                for i in range(self.T + self.standoff_offset, self.T):
                    Q0[:, i] = plans[ind[0]][:, self.T - 1]
        self.solver.reset_initial_seed({f"{self.robot_name}/q/x": self.robot.extract_optimized_dimensions(Q0)})
        solution = self.solver.solve()
        return (solution[f"{self.robot_name}/q"].toarray(), solution[f"{self.robot_name}/dq"].toarray(),
                solution["f"].toarray().flatten())
