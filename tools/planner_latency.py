#!/usr/bin/env python3
"""End-to-end latency of ONE GTOPlanner.plan_goalset call through the drop-in Python surface, the way
examples/pybullet_gto_planning.py:179-190,291-293 calls the reference once per object: scene point cloud -> grid,
cost fields in, one trajectory out (64 candidate grasps as the goal set, the seed chosen among 64 IK solutions).
The reference's published planning_time for this call is 4-30 s.
usage: python tools/planner_latency.py [robot=panda_5k] [n_goals=64] [reps=20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import grasptrajopt_amd as g  # noqa: E402
from grasptrajopt_amd import synthetic as syn  # noqa: E402
from helpers import cfg_of  # noqa: E402


def main():
    robot_name = sys.argv[1] if len(sys.argv) > 1 else "panda_5k"
    n_goals = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    cfg = cfg_of(robot_name)
    robot = g.GTORobotModel(desc=g.load_builtin(robot_name), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=0)
    robot.grid_resolution = 0.0175  # 128^3 over the 2.24 m reach box
    rng = np.random.default_rng(3)
    cloud = rng.uniform([-0.72, -0.72, -0.02], [0.72, 0.72, 1.4], size=(2000, 3))
    robot.setup_points_field(cloud)
    wp = robot.workspace_points
    d_table = wp[:, 2] - 0.0
    q = np.abs(wp - np.array([0.55, 0.1, 0.1])) - np.array([0.06, 0.06, 0.1])
    d_box = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
    # float32, as the fields leave DepthPointCloud.get_sdf_cost (mesh_to_sdf/depth_point_cloud.py:122-141)
    c_all = syn.sdf_cost_map(np.minimum(d_table, d_box), epsilon=0.06).astype(np.float32)
    c_obs = syn.sdf_cost_map(d_table, epsilon=0.06).astype(np.float32)
    planner = g.GTOPlanner(robot, cfg["link_ee"], cfg["link_gripper"], standoff_distance=-0.1, standoff_offset=-10)
    h = robot._util_handle()
    RT, qsol = syn.make_goals(robot.desc, h.eval_fk, cfg["link_ee"], n_goals, seed=11, zlim=(0.15, 0.6))
    qc = np.array(cfg["default_pose"])
    q_solutions = qsol.T.astype(np.float32)
    shape = robot.field_geometry()[0]
    times = []
    for r in range(reps + 2):
        # a new scene every call, as in the driver: perturb the fields so nothing is cached by value
        ca = (c_all + np.float32(1e-6 * r)).astype(np.float32, copy=False)
        t0 = time.perf_counter()
        plan, dQ, cost = planner.plan_goalset(qc, RT, ca, c_obs, [0.0, 0.0, 0.0], q_solutions, use_standoff=True,
                                              axis_standoff=cfg["axis_standoff"], interpolate=True)
        times.append(time.perf_counter() - t0)
    t = np.array(times[2:]) * 1e3
    print(f"{robot_name}: field {tuple(shape)} ({np.prod(shape)} voxels), goal set of {n_goals}, {robot.desc.n_points} surface points")
    print(f"plan_goalset end to end (fields H2D, records + distance fields, seed scoring, solve, D2H): "
          f"median {np.median(t):.2f} ms, min {t.min():.2f}, max {t.max():.2f} over {reps} calls; cost {float(cost[0]):.6f}")
    print(f"iterations of the last solve: {planner.solver.number_of_iterations()}")


if __name__ == "__main__":
    main()
