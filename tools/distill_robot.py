#!/usr/bin/env python3
"""Distil the reference's robot data (URDF + visual meshes + YAML config) into the small
fixtures shipped under grasptrajopt_amd/data/ (SURVEY.md section 2 row 17).

Runs in the build container only (needs /root/reference/data); the GPU box receives just the
resulting <name>.json / <name>.npz / <name>_cfg.json.  Surface points are an *input* of the path
(the reference samples them with an unseeded RNG); here the draw is
seeded and area-weighted on the same visual meshes.
"""
import argparse
import json
import os
import sys

import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grasptrajopt_amd.robot_desc import RobotDesc  # noqa: E402
from grasptrajopt_amd.urdf import Urdf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                   "grasptrajopt_amd", "data"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for robot, variants in (("panda", (("panda", 100), ("panda_5k", 417))), ("fetch", (("fetch", 100),))):
        cfg = yaml.safe_load(open(os.path.join(args.ref, "data", "configs", f"{robot}.yaml")))["robot_cfg"]
        urdf = Urdf.from_file(os.path.join(args.ref, cfg["urdf_robot_path"]))
        model_dir = os.path.join(args.ref, "data", "robots", cfg["robot_name"])
        for name, ppl in variants:
            desc = RobotDesc.from_urdf(urdf, param_joints=cfg["param_joints"],
                                       collision_link_names=cfg["collision_link_names"],
                                       extra_links=[cfg["link_ee"], cfg["link_gripper"]],
                                       model_dir=model_dir, points_per_link=ppl, seed=0)
            desc.save(os.path.join(args.out, name))
            print(name, "frames", desc.n_frames, "links", desc.n_links, "points", desc.n_points,
                  "ndof", desc.ndof, "opt", desc.opt_index.tolist())
        keep = {k: cfg[k] for k in ("robot_name", "link_ee", "link_gripper", "axis_standoff", "default_pose",
                                    "collision_link_names", "param_joints", "finger_index",
                                    "gripper_open_offsets", "arm_len", "arm_height", "retract_distance",
                                    "depth_threshold", "base_link")}
        with open(os.path.join(args.out, f"{robot}_cfg.json"), "w") as fh:
            json.dump(keep, fh, indent=1)


if __name__ == "__main__":
    main()
