"""Shared problem construction for the parity tests."""
import json
import os

import numpy as np

from grasptrajopt_amd import synthetic as syn
from grasptrajopt_amd.robot_desc import RobotDesc, load_builtin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg_of(robot):
    with open(os.path.join(ROOT, "grasptrajopt_amd", "data", f"{robot.split('_')[0]}_cfg.json")) as fh:
        return json.load(fh)


class Problem:
    """A seeded batch of (scene, goal) instances for one robot."""

    def __init__(self, robot="panda", B=4, scene_seed=1, n=48, res=0.0467, n_goals=1, T=50, base=(0, 0, 0),
                 use_standoff=True, fk=None, shelf=False, scene_origin=None, table_z=None):
        self.robot = robot
        self.cfg = cfg_of(robot)
        self.desc = load_builtin(robot)
        self.B, self.T, self.n_goals = B, T, n_goals
        origin = (-0.4, -1.12, -0.4)
        if robot.startswith("fetch"):
            origin = (-0.3, -1.12, 0.0)
        self.scene = syn.make_scene(scene_seed, n=n, res=res, origin=origin if scene_origin is None else scene_origin,
                                    table_z=(0.45 if robot.startswith("fetch") else -0.03) if table_z is None else table_z, shelf=shelf)
        self.base = np.tile(np.asarray(base, dtype=np.float64), (B, 1))
        pose = np.array(self.cfg["default_pose"], dtype=np.float64)
        pose = np.concatenate([np.zeros(self.desc.ndof - len(pose)), pose])  # a planar base in front (robot_desc.with_planar_base)
        self.qc = np.tile(pose, (B, 1))
        self.S = syn.standoff_pose(-0.1, self.cfg["axis_standoff"]) if use_standoff else None
        self._fk = fk

    def finish(self, fk):
        d, cfg, B = self.desc, self.cfg, self.B
        zl = (0.55, 1.2) if self.robot.startswith("fetch") else (0.08, 0.7)
        RT, qg = syn.make_goals(d, fk, cfg["link_ee"], B * self.n_goals, seed=7, zlim=zl)
        self.goals = RT.reshape(B, self.n_goals, 16)
        self.qgoal = qg.reshape(B, self.n_goals, d.ndof)
        self.Q0 = np.stack([syn.make_seed(self.qc[b], self.qgoal[b, 0], self.T, d.param_index) for b in range(B)])
        # parameter joints follow qc
        return self

    def scene_args(self, sid=0):
        sc = self.scene
        return (sid, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)

    def solve_args(self, sid=0):
        return (sid, self.qc, self.goals, self.n_goals, self.S, self.base, self.Q0)


def point_cloud_robot(points):
    """A one-joint 'robot' whose single collision link is the root frame and whose surface points are
    arbitrary query points: gto_eval_points then reports field lookups at exactly these points."""
    P = len(points)
    return RobotDesc(
        name="cloud", frame_names=["root", "tip"], parent=np.array([-1, 0], dtype=np.int32),
        joint_type=np.array([0, 1], dtype=np.int32), q_index=np.array([-1, 0], dtype=np.int32),
        origin_xyz=np.zeros((2, 3)), origin_rpy=np.zeros((2, 3)), axis=np.array([[1.0, 0, 0], [0, 0, 1.0]]),
        actuated_joint_names=["j"], lower=np.array([-1.0]), upper=np.array([1.0]),
        opt_index=np.array([0], dtype=np.int32), param_index=np.array([], dtype=np.int32),
        link_names=["root"], link_frame=np.array([0], dtype=np.int32), visual_xyz=np.zeros((1, 3)),
        visual_rpy=np.zeros((1, 3)), points=np.asarray(points, dtype=np.float64), normals=np.zeros((P, 3)),
        point_link=np.zeros(P, dtype=np.int32))
