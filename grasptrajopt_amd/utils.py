"""Host-side helpers of the GTO layer (gto/utils.py in the reference)."""
from __future__ import annotations

import os

import numpy as np
import yaml

from .synthetic import interpolate_waypoints as _interp2


def get_root_dir() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def load_yaml(file_path):
    """gto/utils.py:15-21."""
    if isinstance(file_path, str):
        with open(file_path) as fh:
            return yaml.load(fh, Loader=yaml.Loader)
    return file_path


def rotZ(rotz: float) -> np.ndarray:
    """gto/utils.py:24-33."""
    c, s = np.cos(rotz), np.sin(rotz)
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])


def interpolate_waypoints(waypoints, n: int, m: int, mode: str = "cubic") -> np.ndarray:
    """gto/utils.py:63-82: clamped cubic through the waypoints sampled at linspace(0,1,n+2)[1:-1]
    (endpoints excluded).  The planner only ever passes two waypoints (gto/gto_planner.py:155,203),
    for which the clamped spline is the closed-form Hermite cubic; more waypoints go through scipy."""
    w = np.asarray(waypoints, dtype=np.float64)
    if mode == "cubic" and w.shape[0] == 2:
        return _interp2(w, n, m)
    from scipy import interpolate
    data = np.zeros([n, m])
    x = np.linspace(0, 1, w.shape[0])
    t = np.linspace(0, 1, n + 2)
    for i in range(w.shape[1]):
        f = (interpolate.interp1d(x, w[:, i], "linear") if mode == "linear"
             else interpolate.CubicSpline(x, w[:, i], bc_type="clamped"))
        data[:, i] = f(t[1:-1])
    return data


def plan_in_collision(robot, depth_pc, plan, base_position=(0.0, 0.0, 0.0), max_points: int = 5, is_mobile: bool = False):
    """The collision statistic of the reference's offline evaluator (examples/pybullet_evaluate_plans.py:
    219-233): a plan collides if at some waypoint more than ``max_points`` robot surface points have a
    negative signed distance to the observed scene.  All waypoints go to the GPU in two calls (FK of the
    surface points, then DepthPointCloud.get_sdf) instead of a Python loop with a KD-tree query per waypoint.
    Returns (in_collision, first_colliding_waypoint or -1, points_in_collision (T,))."""
    plan = np.asarray(plan, dtype=np.float64)
    T = plan.shape[1]
    h = robot._util_handle()
    xyz, _, _, _ = h.eval_points(0, plan.T, [0, 0, 0], want_field=False)  # (T, P, 3), robot-base frame
    if not is_mobile:
        xyz = xyz + np.asarray(base_position, dtype=np.float64).reshape(1, 1, 3)
    sdf = depth_pc.get_sdf(xyz.reshape(-1, 3)).reshape(T, -1)
    count = (sdf < 0).sum(axis=1)
    hit = np.nonzero(count > max_points)[0]
    return bool(hit.size), int(hit[0]) if hit.size else -1, count


def grasp_collision_ratio(gripper_model, depth_pc, RT_grasps, q_gripper, RT_offset=None):
    """The grasp collision filter of the planning driver (examples/pybullet_gto_planning.py:203-219,
    examples/pybullet_gto_planning_mobile.py:307-322): for every candidate grasp pose the open gripper's
    surface points are placed at ``RT_grasp @ RT_offset`` and the fraction with a negative signed distance to
    the observed obstacles is returned (the driver rejects ratios above 0.01).  The gripper's points are
    computed once and all n x P placed points go to the GPU in ONE DepthPointCloud.get_sdf call instead of
    one FK + one KD-tree query per grasp.  Returns ratio (n,) float64."""
    RT = np.asarray(RT_grasps, dtype=np.float64).reshape(-1, 4, 4)
    if RT_offset is not None:
        RT = RT @ np.asarray(RT_offset, dtype=np.float64)
    pts, _ = gripper_model.compute_fk_surface_points(q_gripper)  # gripper frame, (P, 3)
    world = np.einsum("nij,pj->npi", RT[:, :3, :3], pts) + RT[:, None, :3, 3]
    sdf = np.asarray(depth_pc.get_sdf(world.reshape(-1, 3))).reshape(RT.shape[0], -1)
    return (sdf < 0).sum(axis=1) / sdf.shape[1]


def filter_grasps(gripper_model, depth_pc, RT_grasps, q_gripper, RT_offset=None, threshold: float = 0.01):
    """in_collision (n,) int32 exactly as the driver builds it: ratio > threshold."""
    return (grasp_collision_ratio(gripper_model, depth_pc, RT_grasps, q_gripper, RT_offset) > threshold).astype(np.int32)
