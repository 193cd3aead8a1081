"""Flat robot description: the kinematic table + surface points the GTO path consumes.

This is the data the reference assembles from the URDF through ``optas.RobotModel``
(optas/models.py:236-321: joint order, optimised / parameter joints, limits) and
``GTORobotModel`` (gto/gto_models.py:62-101: 100 surface points per collision link, visual
origins).  It is plain arrays so it can cross the C ABI (include/gto_solver.h, gto_robot_desc)
and be committed as a small fixture (JSON + NPZ): the GPU box never sees URDFs or meshes.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from .mesh import load_mesh, sample_surface
from .urdf import Urdf

JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1, 2
_JTYPE = {"fixed": JOINT_FIXED, "revolute": JOINT_REVOLUTE, "continuous": JOINT_REVOLUTE,
          "prismatic": JOINT_PRISMATIC}


@dataclass
class RobotDesc:
    name: str
    # kinematic frames (links), parents before children
    frame_names: List[str]
    parent: np.ndarray       # (F,) int32
    joint_type: np.ndarray   # (F,) int32
    q_index: np.ndarray      # (F,) int32, -1 for fixed
    origin_xyz: np.ndarray   # (F,3)
    origin_rpy: np.ndarray   # (F,3)
    axis: np.ndarray         # (F,3)
    # actuated joints in URDF order (optas/models.py:349-354)
    actuated_joint_names: List[str]
    lower: np.ndarray        # (ndof,)
    upper: np.ndarray        # (ndof,)
    opt_index: np.ndarray    # (n_opt,) int32   (optas/models.py:366-386)
    param_index: np.ndarray  # (n_param,) int32 (optas/models.py:388-396)
    # collision links with surface points (gto/gto_models.py:62-80), URDF link order
    link_names: List[str] = field(default_factory=list)
    link_frame: np.ndarray = None   # (L,) int32
    visual_xyz: np.ndarray = None   # (L,3)
    visual_rpy: np.ndarray = None   # (L,3)
    points: np.ndarray = None       # (P,3) float64, visual-mesh frame
    normals: np.ndarray = None      # (P,3)
    point_link: np.ndarray = None   # (P,) int32

    # ------------------------------------------------------------------ properties
    @property
    def ndof(self) -> int:
        return len(self.actuated_joint_names)

    @property
    def n_opt(self) -> int:
        return int(self.opt_index.shape[0])

    @property
    def n_frames(self) -> int:
        return len(self.frame_names)

    @property
    def n_links(self) -> int:
        return len(self.link_names)

    @property
    def n_points(self) -> int:
        return 0 if self.points is None else int(self.points.shape[0])

    def frame_index(self, link_name: str) -> int:
        try:
            return self.frame_names.index(link_name)
        except ValueError:
            raise KeyError(f"link '{link_name}' is not a kinematic frame of this description "
                           f"(frames: {self.frame_names})") from None

    def link_is_moving(self) -> np.ndarray:
        """(L,) bool: True if an optimised joint lies on the chain from the root to the link."""
        opt = set(self.opt_index.tolist())
        moved = np.zeros(self.n_frames, dtype=bool)
        for i in range(self.n_frames):
            p = self.parent[i]
            moved[i] = (p >= 0 and moved[p]) or (self.q_index[i] in opt)
        return moved[self.link_frame]

    def link_points(self, link_name: str) -> np.ndarray:
        l = self.link_names.index(link_name)
        return self.points[self.point_link == l]

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_urdf(cls, urdf: Urdf, param_joints: Sequence[str] = (),
                  collision_link_names: Optional[Sequence[str]] = None,
                  extra_links: Sequence[str] = (), model_dir: Optional[str] = None,
                  points_per_link: int = 100, seed: int = 0,
                  keep_all_frames: bool = False) -> "RobotDesc":
        actuated = [j.name for j in urdf.joints if j.type != "fixed"]
        for j in urdf.joints:
            if j.type not in _JTYPE:
                raise NotImplementedError(f"{j.type} joints are not supported (optas/models.py:865-866)")
        lower = np.array([urdf.joint_map[j].lower for j in actuated], dtype=np.float64)
        upper = np.array([urdf.joint_map[j].upper for j in actuated], dtype=np.float64)
        params = [j for j in actuated if j in set(param_joints)]
        opt_index = np.array([i for i, j in enumerate(actuated) if j not in params], dtype=np.int32)
        param_index = np.array([i for i, j in enumerate(actuated) if j in params], dtype=np.int32)

        # collision links in URDF link order, as the reference's surface_pc_map is keyed
        links = [l for l in urdf.links if l.has_visual and l.visual_mesh is not None
                 and (collision_link_names is None or l.name in collision_link_names)]

        root = urdf.get_root()
        needed = set()
        targets = [l.name for l in links] + list(extra_links)
        if keep_all_frames:
            targets = [l.name for l in urdf.links]
        for name in targets:
            if name not in urdf.link_map:
                raise KeyError(f"link '{name}' does not appear in URDF")
            cur = name
            while True:
                needed.add(cur)
                if cur == root:
                    break
                cur = urdf.parent_joint[cur].parent
        children: Dict[str, List[str]] = {}
        for j in urdf.joints:
            children.setdefault(j.parent, []).append(j.child)
        order: List[str] = []
        stack = [root]
        while stack:  # depth-first, document order of joints
            cur = stack.pop()
            if cur not in needed:
                continue
            order.append(cur)
            stack.extend(reversed(children.get(cur, [])))
        fidx = {n: i for i, n in enumerate(order)}
        F = len(order)
        parent = np.full(F, -1, dtype=np.int32)
        jtype = np.zeros(F, dtype=np.int32)
        qidx = np.full(F, -1, dtype=np.int32)
        oxyz = np.zeros((F, 3))
        orpy = np.zeros((F, 3))
        axis = np.tile(np.array([1.0, 0.0, 0.0]), (F, 1))
        for i, name in enumerate(order):
            if name == root:
                continue
            j = urdf.parent_joint[name]
            parent[i] = fidx[j.parent]
            jtype[i] = _JTYPE[j.type]
            if j.type != "fixed":
                qidx[i] = actuated.index(j.name)
            oxyz[i] = j.xyz
            orpy[i] = j.rpy
            axis[i] = j.axis

        desc = cls(name=urdf.name, frame_names=order, parent=parent, joint_type=jtype, q_index=qidx,
                   origin_xyz=oxyz, origin_rpy=orpy, axis=axis, actuated_joint_names=actuated,
                   lower=lower, upper=upper, opt_index=opt_index, param_index=param_index)
        desc.link_names = [l.name for l in links]
        desc.link_frame = np.array([fidx[l.name] for l in links], dtype=np.int32)
        desc.visual_xyz = np.array([l.visual_xyz for l in links], dtype=np.float64).reshape(-1, 3)
        desc.visual_rpy = np.array([l.visual_rpy for l in links], dtype=np.float64).reshape(-1, 3)
        if model_dir is not None and links:
            pts, nrm, pl = [], [], []
            for li, l in enumerate(links):
                v, f = load_mesh(os.path.join(model_dir, l.visual_mesh))
                v = v * np.asarray(l.visual_scale, dtype=np.float64)[None, :]
                p, nn = sample_surface(v, f, points_per_link, seed=seed + li)
                pts.append(p)
                nrm.append(nn)
                pl.append(np.full(points_per_link, li, dtype=np.int32))
            desc.points = np.concatenate(pts)
            desc.normals = np.concatenate(nrm)
            desc.point_link = np.concatenate(pl)
        return desc

    # ------------------------------------------------------------------ fixtures
    def save(self, prefix: str) -> None:
        meta = dict(name=self.name, frame_names=self.frame_names,
                    actuated_joint_names=self.actuated_joint_names, link_names=self.link_names)
        with open(prefix + ".json", "w") as fh:
            json.dump(meta, fh, indent=1)
        np.savez_compressed(
            prefix + ".npz", parent=self.parent, joint_type=self.joint_type, q_index=self.q_index,
            origin_xyz=self.origin_xyz, origin_rpy=self.origin_rpy, axis=self.axis, lower=self.lower,
            upper=self.upper, opt_index=self.opt_index, param_index=self.param_index,
            link_frame=self.link_frame, visual_xyz=self.visual_xyz, visual_rpy=self.visual_rpy,
            points=self.points, normals=self.normals.astype(np.float32), point_link=self.point_link)

    @classmethod
    def load(cls, prefix: str) -> "RobotDesc":
        with open(prefix + ".json") as fh:
            meta = json.load(fh)
        z = np.load(prefix + ".npz")
        return cls(name=meta["name"], frame_names=meta["frame_names"], parent=z["parent"],
                   joint_type=z["joint_type"], q_index=z["q_index"], origin_xyz=z["origin_xyz"],
                   origin_rpy=z["origin_rpy"], axis=z["axis"],
                   actuated_joint_names=meta["actuated_joint_names"], lower=z["lower"], upper=z["upper"],
                   opt_index=z["opt_index"], param_index=z["param_index"], link_names=meta["link_names"],
                   link_frame=z["link_frame"], visual_xyz=z["visual_xyz"], visual_rpy=z["visual_rpy"],
                   points=z["points"], normals=z["normals"].astype(np.float64), point_link=z["point_link"])


def with_planar_base(desc: RobotDesc, xy_limit: float = 1.0, name: Optional[str] = None) -> RobotDesc:
    """The same robot on a planar mobile base: three actuated joints (prismatic x, prismatic y, revolute about z) between a
    fixed world frame and the old root, optimised together with the arm -- the mobile manipulator of
    examples/pybullet_gto_planning_mobile.py with the base pose as part of the trajectory (BASELINE configs[4]: Fetch
    arm 7 + base 3 = 10 optimised joints).  Joint order: the base joints first, then the robot's own."""
    F0 = desc.n_frames
    names = ["odom", "base_x", "base_y", "base_theta"] + list(desc.frame_names)
    parent = np.concatenate([[-1, 0, 1, 2], np.where(desc.parent < 0, 3, desc.parent + 4)]).astype(np.int32)
    jtype = np.concatenate([[JOINT_FIXED, JOINT_PRISMATIC, JOINT_PRISMATIC, JOINT_REVOLUTE], desc.joint_type]).astype(np.int32)
    qidx = np.concatenate([[-1, 0, 1, 2], np.where(desc.q_index < 0, -1, desc.q_index + 3)]).astype(np.int32)
    z3 = np.zeros((4, 3))
    axis = np.concatenate([[[1.0, 0, 0], [1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]], desc.axis])
    out = RobotDesc(
        name=name or desc.name + "_mobile", frame_names=names, parent=parent, joint_type=jtype, q_index=qidx,
        origin_xyz=np.concatenate([z3, desc.origin_xyz]), origin_rpy=np.concatenate([z3, desc.origin_rpy]), axis=axis,
        actuated_joint_names=["base_x", "base_y", "base_theta"] + list(desc.actuated_joint_names),
        lower=np.concatenate([[-xy_limit, -xy_limit, -np.pi], desc.lower]), upper=np.concatenate([[xy_limit, xy_limit, np.pi], desc.upper]),
        opt_index=np.concatenate([[0, 1, 2], desc.opt_index + 3]).astype(np.int32), param_index=(desc.param_index + 3).astype(np.int32),
        link_names=list(desc.link_names), link_frame=(desc.link_frame + 4).astype(np.int32), visual_xyz=desc.visual_xyz.copy(),
        visual_rpy=desc.visual_rpy.copy(), points=desc.points.copy(), normals=desc.normals.copy(), point_link=desc.point_link.copy())
    assert out.n_frames == F0 + 4
    return out


_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_builtin(name: str) -> RobotDesc:
    """Distilled fixtures shipped with the package (tools/distill_robot.py): 'panda', 'fetch',
    'panda_5k' (12 x 417 points, the benchmark's ~5k-point robot)."""
    if name.endswith("_mobile"):  # 'fetch_mobile': the Fetch arm on a planar base, 10 optimised joints
        return with_planar_base(load_builtin(name[: -len("_mobile")]), name=name)
    prefix = os.path.join(_DATA_DIR, name)
    if not os.path.exists(prefix + ".npz"):
        raise FileNotFoundError(f"no built-in robot description '{name}' under {_DATA_DIR}")
    return RobotDesc.load(prefix)
