#!/usr/bin/env python3
"""Per-object latency of the whole drop-in pipeline, stage by stage, through the Python surface that mirrors
examples/pybullet_gto_planning.py:
  depth image -> two DepthPointCloud objects, grid from the first one's points (:176-179) -> cost fields sdf_cost_all /
  sdf_cost_obstacle (:181-190, two get_sdf_cost calls) -> IK pre-filter of the candidate grasps (:242-272, one
  solve_ik_batch instead of a loop of IPOPT runs) -> plan_goalset (:291).
Since round 4 the first two stages only hand out lazy stand-ins (grasptrajopt_amd/depth_scene.py); the GPU work of the
perception steps happens in ONE gto_scene_from_depth call when the first consumer needs a scene, timed here as its own
stage (`resident scene`).  `host=True` forces the round-3 path (points, grid and fields through numpy) for comparison.
The reference reports 1.6-2.5 s for the IK loop and 4-30 s planning_time per object; its KD-tree field takes seconds.
usage: python tools/pipeline_latency.py [grid_resolution=0.05] [n_grasps=64] [reps=10]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stage_table(res=0.05, n_goals=64, reps=10, host=False, device=0):
    import grasptrajopt_amd as g
    from grasptrajopt_amd import synthetic as syn
    from helpers import cfg_of
    cfg = cfg_of("panda_5k")
    robot = g.GTORobotModel(desc=g.load_builtin("panda_5k"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=device)
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
        pts = dpc_all.points
        robot.setup_points_field(np.asarray(pts) if host else pts)
        t.append(time.perf_counter())
        c_all = dpc_all.get_sdf_cost(robot.workspace_points)
        c_obs = dpc_obs.get_sdf_cost(robot.workspace_points)
        t.append(time.perf_counter())
        if hasattr(c_obs, "ensure_scene"):
            c_obs.ensure_scene()  # (the IK call would do it: timed as a stage of its own)
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
    out = {"grid_resolution_m": res, "field_shape": [int(x) for x in shape], "voxels": int(np.prod(shape)), "candidate_grasps": n_goals,
           "grasps_passing_ik": int(ok.sum()), "image": f"{H}x{W} depth", "path": "host (numpy points, grid and fields: rounds 2-3)" if host else "device-resident (depth_scene.py)",
           "ms": {"clouds_and_grid": round(float(ms[0]), 3), "two_cost_fields": round(float(ms[1]), 3), "resident_scene_build": round(float(ms[2]), 3),
                  "ik_of_all_grasps": round(float(ms[3]), 3), "plan_goalset": round(float(ms[4]), 3), "per_object": round(float(ms.sum()), 3)},
           "plan_cost": float(cost[0]), "reps": reps}
    robot.close()
    return out


def main():
    res = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
    n_goals = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    for host in (False, True):
        o = stage_table(res, n_goals, reps, host)
        m = o["ms"]
        print(f"[{o['path']}] grid {res*100:.2f} cm -> field {tuple(o['field_shape'])} ({o['voxels']} voxels), {n_goals} candidate grasps "
              f"({o['grasps_passing_ik']} pass the IK thresholds), {o['image']}")
        print(f"  point clouds + grid      {m['clouds_and_grid']:7.2f} ms")
        print(f"  two cost fields          {m['two_cost_fields']:7.2f} ms")
        print(f"  resident scene build     {m['resident_scene_build']:7.2f} ms")
        print(f"  IK of all grasps         {m['ik_of_all_grasps']:7.2f} ms")
        print(f"  plan_goalset             {m['plan_goalset']:7.2f} ms")
        print(f"  per object, end to end   {m['per_object']:7.2f} ms (median of {reps}); plan cost {o['plan_cost']:.6f}")


if __name__ == "__main__":
    main()
