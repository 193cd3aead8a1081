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


def random_robot(seed, n_frames=None, n_opt=None):
    """A random kinematic tree with surface points: branching, revolute / prismatic / fixed joints in any order, random
    origins and axes, optimised joints along one chain (the one that carries the end effector) and parameter joints on
    side branches, collision links on some frames.  Exercises what the built-in arms do not: branching before and after
    optimised joints, prismatic optimised joints, fixed frames inside the chain, up to 24 frames."""
    rng = np.random.default_rng(seed)
    F = int(n_frames or rng.integers(8, 25))
    n = int(n_opt or rng.integers(3, 9))
    chain_len = int(min(F - 1, n + rng.integers(0, 4)))   # frames of the main chain below the root (some fixed)
    parent = np.full(F, -1, dtype=np.int32)
    jt = np.zeros(F, dtype=np.int32)
    chain = list(range(1, chain_len + 1))
    for k, f in enumerate(chain):
        parent[f] = f - 1
    movable = rng.permutation(chain)[:n]
    for f in movable:
        jt[f] = 2 if rng.random() < 0.25 else 1
    side = list(range(chain_len + 1, F))
    for f in side:                                       # side branches hang anywhere on what exists already
        parent[f] = int(rng.integers(0, f))
        jt[f] = int(rng.choice([0, 1, 2], p=[0.4, 0.4, 0.2]))
    act = [f for f in range(F) if jt[f] != 0]
    q_index = np.full(F, -1, dtype=np.int32)
    for k, f in enumerate(act):
        q_index[f] = k
    ndof = len(act)
    opt_index = np.array(sorted(q_index[f] for f in movable), dtype=np.int32)
    param_index = np.array([k for k in range(ndof) if k not in set(opt_index.tolist())], dtype=np.int32)
    origin_xyz = rng.uniform(-0.12, 0.12, size=(F, 3))
    origin_xyz[chain, 2] = rng.uniform(0.08, 0.22, size=len(chain))   # the chain reaches outward
    origin_rpy = rng.uniform(-1.5, 1.5, size=(F, 3))
    origin_xyz[0] = origin_rpy[0] = 0.0
    axis = rng.standard_normal((F, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    lower = np.where([jt[f] == 2 for f in act], -0.15, -2.2) * rng.uniform(0.6, 1.0, ndof)
    upper = np.where([jt[f] == 2 for f in act], 0.15, 2.2) * rng.uniform(0.6, 1.0, ndof)
    link_frames = sorted(set([0, chain[-1]] + [int(f) for f in rng.choice(np.arange(1, F), size=min(F - 1, int(rng.integers(3, 9))), replace=False)]))
    L = len(link_frames)
    pts, plink = [], []
    for l in range(L):
        m = int(rng.integers(40, 160))
        c = rng.uniform(-0.03, 0.03, 3)
        pts.append(c + rng.standard_normal((m, 3)) * rng.uniform(0.015, 0.05, 3))
        plink.append(np.full(m, l, dtype=np.int32))
    names = [f"f{i}" for i in range(F)]
    return RobotDesc(
        name=f"random{seed}", frame_names=names, parent=parent, joint_type=jt, q_index=q_index, origin_xyz=origin_xyz,
        origin_rpy=origin_rpy, axis=axis, actuated_joint_names=[f"j{k}" for k in range(ndof)], lower=lower, upper=upper,
        opt_index=opt_index, param_index=param_index, link_names=[names[f] for f in link_frames],
        link_frame=np.array(link_frames, dtype=np.int32), visual_xyz=rng.uniform(-0.02, 0.02, (L, 3)),
        visual_rpy=rng.uniform(-0.5, 0.5, (L, 3)), points=np.concatenate(pts), normals=np.zeros((sum(len(p) for p in pts), 3)),
        point_link=np.concatenate(plink)), names[chain[-1]]
