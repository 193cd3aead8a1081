"""Result files of the planning driver and the statistics of the offline evaluator.

Wire format of examples/results_iros2024/*.json, written by examples/pybullet_gto_planning.py:323-338:
    {scene_id: {ordering: {object_name: {"reward": int, "plan": [[...]*T]*ndof | null,
                                         "checking_time": s | null, "ik_time": s | null, "planning_time": s | null}}}}
(the mobile driver adds "RT_base_new": 4x4 list next to the objects of an ordering,
examples/pybullet_gto_planning_mobile.py:257).  Statistics as examples/pybullet_evaluate_plans.py:150-290
prints them: per-object trial / success / collision counts and the mean checking, IK and planning times over
the trials that have them.
"""
from __future__ import annotations

import datetime
import json
import os
from typing import Callable, Dict, Iterator, Optional, Tuple

import numpy as np

TIME_KEYS = ("checking_time", "ik_time", "planning_time")


def object_result(reward, plan, checking_time=None, ik_time=None, planning_time=None) -> dict:
    """One entry of the result tree (examples/pybullet_gto_planning.py:321-322; failures carry plan None)."""
    return {"reward": int(reward), "plan": None if plan is None else np.asarray(plan).tolist(),
            "checking_time": checking_time, "ik_time": ik_time, "planning_time": planning_time}


def result_filename(robot_name: str, scene_type: str, planner: str = "GTO", mobile: bool = False,
                    now: Optional[datetime.datetime] = None) -> str:
    """GTO_scenereplica_{robot}_{scene_type}_{yy-mm-dd_THHMMSS}.json (:335-336; 'mobile_' prefix in the mobile driver)."""
    stamp = "{:%y-%m-%d_T%H%M%S}".format(now or datetime.datetime.now())
    return f"{planner}_scenereplica_{'mobile_' if mobile else ''}{robot_name}_{scene_type}_{stamp}.json"


def write_results(results_scene: dict, outdir: str, robot_name: str, scene_type: str, planner: str = "GTO",
                  mobile: bool = False, now: Optional[datetime.datetime] = None) -> str:
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, result_filename(robot_name, scene_type, planner, mobile, now))
    with open(path, "w") as fh:
        json.dump(results_scene, fh)
    return path


def read_results(path: str) -> dict:
    with open(path) as fh:
        return json.load(fh)


def iter_trials(results_scene: dict) -> Iterator[Tuple[str, str, str, dict]]:
    """(scene_id, ordering, object_name, entry) in file order; non-object keys (RT_base_new) are skipped."""
    for scene_id, orderings in results_scene.items():
        for ordering, objects in orderings.items():
            for name, entry in objects.items():
                if isinstance(entry, dict) and "reward" in entry:
                    yield scene_id, ordering, name, entry


def plan_array(entry: dict) -> Optional[np.ndarray]:
    """(ndof, T) float64 or None."""
    return None if entry.get("plan") is None else np.asarray(entry["plan"], dtype=np.float64)


def summarize(results_scene: dict, in_collision: Optional[Callable[[str, str, str, np.ndarray], bool]] = None) -> dict:
    """The evaluator's totals (examples/pybullet_evaluate_plans.py:162-181,236-268).  ``in_collision(scene,
    ordering, object, plan)`` (e.g. built on utils.plan_in_collision) is asked for every stored plan."""
    per_object: Dict[str, Dict[str, int]] = {}
    sums = {k: 0.0 for k in TIME_KEYS}
    counts = {k: 0 for k in TIME_KEYS}
    total_success = total_collision = trials = 0
    for scene_id, ordering, name, entry in iter_trials(results_scene):
        o = per_object.setdefault(name, {"total": 0, "success": 0, "collision": 0})
        o["total"] += 1
        o["success"] += int(entry["reward"])
        total_success += int(entry["reward"])
        trials += 1
        for k in TIME_KEYS:
            if entry.get(k) is not None:  # checking_time may be absent in older files (:171-175)
                sums[k] += entry[k]
                counts[k] += 1
        if in_collision is not None and entry.get("plan") is not None:
            hit = bool(in_collision(scene_id, ordering, name, plan_array(entry)))
            o["collision"] += int(hit)
            total_collision += int(hit)
    mean = {k: (sums[k] / counts[k] if counts[k] else None) for k in TIME_KEYS}
    return {"per_object": per_object, "total_success": total_success, "total_collision": total_collision,
            "total_trial": trials, "mean_time": mean, "total_time": sum(v for v in mean.values() if v is not None)}


def plan_shape_statistics(plans, opt_index, lower, upper, standoff_waypoint=None) -> Dict[str, np.ndarray]:
    """Per-plan statistics of joint trajectories `plans` (n, ndof, T) that need nothing but the plan itself -- what can be
    compared between the reference's stored plans (examples/results_iros2024/*.json: outputs without their inputs) and plans
    solved here on other inputs (tests/test_plan_statistics_cpu.py).  The reference's objective
    (gto/gto_planner.py:84-135) has a velocity term, a goal term at the last waypoint and a standoff term at waypoint
    T + standoff_offset; its minimiser in joint space is two constant-speed stretches joined at the standoff waypoint.
      v0           speed of the first step over the plan's mean speed (initial velocity is constrained to zero)
      cv_pre/post  coefficient of variation of the step lengths before / after the standoff waypoint (0: constant speed)
      plateau      mean step length before over mean step length after the standoff waypoint
      pre_share    share of the joint-space path length covered before the standoff waypoint
      path_ratio   joint-space path length over the distance between the first and the last configuration (>= 1)
      chord_dev    largest distance of the waypoints before the standoff waypoint from the straight line between the first free
                   waypoint and the standoff waypoint, over that line's length (0: a straight line in joint space)
      on_bound     share of waypoints with an optimised joint within 1e-6 of a limit
      chord        joint-space distance between the first and the last configuration (rad; what a workload's goals have to
                   look like to be "reference-shaped": bench.py's quality.reference_shaped)"""
    P = np.asarray(plans, dtype=np.float64)
    o = np.asarray(opt_index)
    Q = P[:, o, :]
    T = Q.shape[2]
    ts = T - 10 if standoff_waypoint is None else int(standoff_waypoint)  # reference: standoff_offset = -10
    step = np.linalg.norm(np.diff(Q, axis=2), axis=1)  # (n, T - 1)
    tiny = 1e-300
    pre, post = step[:, 1:ts - 1], step[:, ts + 1:T - 2]
    L = step.sum(axis=1)
    a, b = Q[:, :, 1], Q[:, :, ts]
    tt = np.linspace(0.0, 1.0, ts)
    chord = a[:, :, None] + (b - a)[:, :, None] * tt[None, None, :]
    lo, hi = np.asarray(lower)[None, :, None], np.asarray(upper)[None, :, None]
    return {
        "v0": step[:, 0] / np.maximum(step.mean(axis=1), tiny),
        "cv_pre": pre.std(axis=1) / np.maximum(pre.mean(axis=1), tiny),
        "cv_post": post.std(axis=1) / np.maximum(post.mean(axis=1), tiny),
        "plateau": pre.mean(axis=1) / np.maximum(post.mean(axis=1), tiny),
        "pre_share": step[:, :ts].sum(axis=1) / np.maximum(L, tiny),
        "path_ratio": L / np.maximum(np.linalg.norm(Q[:, :, -1] - Q[:, :, 0], axis=1), tiny),
        "chord_dev": np.linalg.norm(Q[:, :, 1:ts + 1] - chord, axis=1).max(axis=1) / np.maximum(np.linalg.norm(b - a, axis=1), tiny),
        "on_bound": ((np.abs(Q - lo) <= 1e-6) | (np.abs(Q - hi) <= 1e-6)).any(axis=1).mean(axis=1),
        "chord": np.linalg.norm(Q[:, :, -1] - Q[:, :, 0], axis=1),
    }
