#!/usr/bin/env python3
"""Per-object latency of the whole drop-in pipeline, stage by stage, through the Python surface that mirrors
examples/pybullet_gto_planning.py:
  depth image -> point cloud -> grid (:176-179) -> cost fields sdf_cost_all / sdf_cost_obstacle (:181-184, two
  DepthPointCloud.get_sdf_cost calls) -> IK pre-filter of the candidate grasps (:242-272, one solve_ik_batch instead of
  a loop of IPOPT runs) -> plan_goalset (:291).
The reference reports 1.6-2.5 s for the IK loop and 4-30 s planning_time per object; its KD-tree field takes seconds.
usage: python tools/pipeline_latency.py [grid_resolution=0.05] [n_grasps=64] [reps=10]"""
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
    res = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
    n_goals = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    cfg = cfg_of("panda_5k")
    robot = g.GTORobotModel(desc=g.load_builtin("panda_5k"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=0)
    robot.grid_resolution = res
    # a camera looking down at a table with two boxes, 480 x 640
    H, W = 480, 640
    K = np.array([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1.0]])
    v, u = np.mgrid[0:H, 0:W]
    depth = (1.0 + 0.0012 * (v - H / 2) + 0.0003 * (u - W / 2)).astype(np.float32)
    for (r0, r1, c0, c1, dz) in ((150, 260, 200, 330, 0.2), (280, 400, 380, 520, 0.1)):
        depth[r0:r1, c0:c1] -= dz
    target = np.zeros((H, W), np.uint8)
    target[150:260, 200:330] = 1  # the object to grasp is left out of sdf_cost_obstacle
    a = 0.9
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, -np.sin(a), np.cos(a)], [-1.0, 0, 0], [0, -np.cos(a), -np.sin(a)]])
    cam[:3, 3] = [-0.1, 0.0, 0.9]
    planner = g.GTOPlanner(robot, cfg["link_ee"], cfg["link_gripper"], standoff_distance=-0.1, standoff_offset=-10)
    ik = g.IKSolver(robot, cfg["link_ee"], cfg["link_gripper"], collision_avoidance=True)
    RT, _ = syn.make_goals(robot.desc, robot._util_handle().eval_fk, cfg["link_ee"], n_goals, seed=11, zlim=(0.15, 0.6))
    qc = np.array(cfg["default_pose"])
    rows = []
    for r in range(reps + 2):
        d = depth + np.float32(1e-4 * r)
        t = [time.perf_counter()]
        dpc_all = g.DepthPointCloud(d, K, cam, target_mask=None, threshold=2.0)
        dpc_obs = g.DepthPointCloud(d, K, cam, target_mask=target, threshold=2.0)
        robot.setup_points_field(dpc_all.points)
        t.append(time.perf_counter())
        c_all = dpc_all.get_sdf_cost(robot.workspace_points)
        c_obs = dpc_obs.get_sdf_cost(robot.workspace_points)
        t.append(time.perf_counter())
        q_ik, ep, er, cost_ik, it, st = ik.solve_ik_batch(qc, RT, c_obs, [0.0, 0.0, 0.0])
        ok = (ep < 0.01) & (er < 5.0)
        t.append(time.perf_counter())
        sel = np.nonzero(ok)[0] if ok.any() else np.arange(n_goals)
        plan, dQ, cost = planner.plan_goalset(qc, RT[sel], c_all, c_obs, [0.0, 0.0, 0.0], q_ik[sel].T.astype(np.float32),
                                              use_standoff=True, axis_standoff=cfg["axis_standoff"], interpolate=True)
        t.append(time.perf_counter())
        rows.append(np.diff(t))
    ms = np.median(np.array(rows[2:]), axis=0) * 1e3
    shape = robot.field_geometry()[0]
    print(f"grid {res*100:.2f} cm -> field {tuple(shape)} ({int(np.prod(shape))} voxels), {n_goals} candidate grasps ({int(ok.sum())} pass the IK thresholds), 480x640 depth")
    print(f"  point clouds + grid      {ms[0]:7.2f} ms")
    print(f"  two cost fields          {ms[1]:7.2f} ms")
    print(f"  IK of all grasps         {ms[2]:7.2f} ms")
    print(f"  plan_goalset             {ms[3]:7.2f} ms")
    print(f"  per object, end to end   {ms.sum():7.2f} ms (median of {reps})")


if __name__ == "__main__":
    main()
