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
