"""GTORobotModel — the robot-model surface the reference's drivers use (gto/gto_models.py:23-215),
backed by the flat RobotDesc and the HIP library.  No CasADi, urdf_parser_py, trimesh or sklearn."""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _capi
from .optas_facade import DM
from .robot_desc import RobotDesc
from .urdf import Urdf


class GTORobotModel:
    def __init__(self, model_dir: Optional[str] = None, urdf_filename: Optional[str] = None,
                 urdf_string: Optional[str] = None, xacro_filename: Optional[str] = None,
                 name: Optional[str] = None, time_derivs: Sequence[int] = (0,), qddlim=None, T=None,
                 param_joints: Sequence[str] = (), collision_link_names: Optional[Sequence[str]] = None,
                 points_per_link: int = 100, seed: int = 0, desc: Optional[RobotDesc] = None, device: int = -1):
        """Same arguments as the reference (gto/gto_models.py:26-46).  ``desc`` short-circuits URDF and
        mesh parsing with a pre-distilled description (robot_desc.load_builtin)."""
        if xacro_filename is not None:
            raise NotImplementedError("xacro input is not supported; pass a URDF")
        if desc is None:
            if urdf_filename is not None:
                urdf = Urdf.from_file(urdf_filename)
            elif urdf_string is not None:
                urdf = Urdf.from_string(urdf_string)
            else:
                raise AssertionError("You need to supply a urdf, either through filename or as a string")
            if model_dir is None:
                model_dir = os.path.dirname(urdf_filename) if urdf_filename else "."
            desc = RobotDesc.from_urdf(urdf, param_joints=param_joints, collision_link_names=collision_link_names,
                                       model_dir=model_dir, points_per_link=points_per_link, seed=seed,
                                       keep_all_frames=len(urdf.links) <= _capi.GTO_MAX_FRAMES)
        self.desc = desc
        self.model_dir = model_dir
        self.name = name or desc.name
        self.time_derivs = list(time_derivs)
        self.param_joints = list(param_joints)
        self.collision_link_names = collision_link_names
        self.field_margin = 0.4        # gto/gto_models.py:45
        self.grid_resolution = 0.05    # gto/gto_models.py:46
        self.device = device
        self._handles: Dict[tuple, _capi.SolverHandle] = {}
        # surface_pc_map[name].points / .normals (gto/gto_models.py:62-80)
        self.surface_pc_map = {
            ln: SimpleNamespace(points=desc.points[desc.point_link == i], normals=desc.normals[desc.point_link == i])
            for i, ln in enumerate(desc.link_names)}

    # ------------------------------------------------------------------ optas.RobotModel surface
    def get_name(self) -> str:
        return self.name

    @property
    def ndof(self) -> int:
        return self.desc.ndof

    @property
    def link_names(self) -> List[str]:
        return list(self.desc.frame_names)

    @property
    def actuated_joint_names(self) -> List[str]:
        return list(self.desc.actuated_joint_names)

    @property
    def optimized_joint_indexes(self) -> List[int]:
        return self.desc.opt_index.tolist()

    @property
    def parameter_joint_indexes(self) -> List[int]:
        return self.desc.param_index.tolist()

    @property
    def optimized_joint_names(self) -> List[str]:
        return [self.desc.actuated_joint_names[i] for i in self.desc.opt_index]

    @property
    def parameter_joint_names(self) -> List[str]:
        return [self.desc.actuated_joint_names[i] for i in self.desc.param_index]

    @property
    def num_opt_joints(self) -> int:
        return self.desc.n_opt

    @property
    def num_param_joints(self) -> int:
        return int(self.desc.param_index.shape[0])

    @property
    def lower_actuated_joint_limits(self) -> DM:
        return DM(self.desc.lower)

    @property
    def upper_actuated_joint_limits(self) -> DM:
        return DM(self.desc.upper)

    @property
    def lower_optimized_joint_limits(self) -> DM:
        return DM(self.desc.lower[self.desc.opt_index])

    @property
    def upper_optimized_joint_limits(self) -> DM:
        return DM(self.desc.upper[self.desc.opt_index])

    def extract_parameter_dimensions(self, values):
        return np.asarray(values)[self.parameter_joint_indexes, :]

    def extract_optimized_dimensions(self, values):
        return np.asarray(values)[self.optimized_joint_indexes, :]

    # ------------------------------------------------------------------ HIP handles
    def solver_handle(self, link_ee: str, link_gripper: str, opts=None, role: str = "planner") -> _capi.SolverHandle:
        """One gto_handle per (role, link_ee, link_gripper, T, standoff_offset); created on first use.  The role
        ("planner", "ik", "base") keeps the planner's mutable options (weights, iteration cap) and scenes away
        from the IK and base solvers, whose weights are the reference's constants (gto/ik_solver.py:70)."""
        o = opts if opts is not None else _capi.default_opts()
        key = (role, link_ee, link_gripper, int(o.T), int(o.standoff_offset))
        h = self._handles.get(key)
        if h is None:
            h = _capi.SolverHandle(self.desc, link_ee, link_gripper, o, device=self.device)
            self._handles[key] = h
        return h

    SCRATCH_SCENE = 1  # scene id of one-off evaluations (the planner's own scene is 0)

    def _util_handle(self) -> _capi.SolverHandle:
        """Handle of the stand-alone evaluations (FK, surface points, plan cost): its own role, so that a scratch
        scene never replaces one a solver is using."""
        ln = self.desc.link_names[-1]
        return self.solver_handle(ln, ln, role="util")

    # ------------------------------------------------------------------ FK of the surface points
    def compute_fk_surface_points(self, q_user_input, tf_base=None):
        """gto/gto_models.py:104-121 -> (points (M,3), normals (M,3)) in the base (or tf_base) frame."""
        h = self._util_handle()
        q = np.asarray(q_user_input, dtype=np.float64).reshape(1, self.ndof)
        xyz, _, _, _ = h.eval_points(0, q, [0, 0, 0], want_field=False)
        frames = h.eval_fk(q)[0]
        d = self.desc
        nrm = np.empty_like(d.normals)
        for l in range(d.n_links):
            R = frames[d.link_frame[l]][:3, :3] @ _rpy2r(d.visual_rpy[l])
            sel = d.point_link == l
            nrm[sel] = d.normals[sel] @ R.T
        pts = xyz[0]
        if tf_base is not None:
            tf_base = np.asarray(tf_base, dtype=np.float64)
            pts = pts @ tf_base[:3, :3].T + tf_base[:3, 3]
            nrm = nrm @ tf_base[:3, :3].T
        return pts, nrm

    def compute_fk_link_surface_points(self, q_user_input, name, tf_base=None):
        pts, _ = self.compute_fk_surface_points(q_user_input, tf_base)
        return pts[self.desc.point_link == self.desc.link_names.index(name)]

    # ------------------------------------------------------------------ voxel grid (gto/gto_models.py:135-201)
    def _setup_field(self, lo, hi):
        self._pending_depth = None  # a grid given as numbers: no depth cloud is waiting to size one
        m, r = self.field_margin, self.grid_resolution
        self.origin = np.array([lo[0] - m, lo[1] - m, lo[2] - m]).reshape((1, 3))
        axes = [np.arange(lo[a] - m, hi[a] + m, r) for a in range(3)]
        wp = np.array(np.meshgrid(*axes, indexing="ij"))
        self.field_shape = wp.shape[1:]
        self.workspace_points = wp.reshape((3, -1)).T
        self.field_size = self.workspace_points.shape[0]

    def setup_workspace_field(self, arm_len, arm_height):
        self.xlim, self.ylim, self.zlim = [0, arm_len], [-arm_len, arm_len], [0, arm_height + arm_len]
        self._setup_field([self.xlim[0], self.ylim[0], self.zlim[0]], [self.xlim[1], self.ylim[1], self.zlim[1]])

    _FIELD_ATTRS = ("origin", "field_shape", "workspace_points", "field_size", "workspace_bounds")

    def setup_points_field(self, points):
        from .depth_scene import LazyCloudPoints
        if isinstance(points, LazyCloudPoints) and points.dpc.target_mask is None:
            # the cloud is still on the device: the grid is sized there, when the first consumer needs a scene
            # (depth_scene.py); until then the geometry attributes are pending.  (The resident build sizes the grid from the
            # cloud of ALL pixels, as the driver does: a masked cloud takes the host path below.)
            for a in self._FIELD_ATTRS:
                self.__dict__.pop(a, None)
            self._pending_depth = points.dpc
            return
        points = np.asarray(points)
        self.workspace_bounds = np.stack((points.min(0), points.max(0)), axis=1)
        self._setup_field(self.workspace_bounds[:, 0], self.workspace_bounds[:, 1])

    def __getattr__(self, name):
        # only reached for attributes that are not set: the grid geometry while a depth cloud is pending
        if name in GTORobotModel._FIELD_ATTRS and self.__dict__.get("_pending_depth") is not None:
            if name == "workspace_points":
                from .depth_scene import LazyWorkspacePoints
                return LazyWorkspacePoints(self)
            self._resolve_depth_field()
            return self.__dict__[name]
        raise AttributeError(f"{type(self).__name__!s} object has no attribute {name!r}")

    def _resolve_depth_field(self):
        """Geometry of the pending depth grid: from the resident scene if a consumer has built one, else by building it
        from the cloud of all pixels."""
        if "field_shape" not in self.__dict__:
            pend = self.__dict__.get("_pending_depth")
            if pend is None:
                raise RuntimeError("call setup_points_field() or setup_workspace_field() first")
            self._depth_scene_for(pend, 0.02, 1.0, pend)

    def _workspace_points_now(self):
        """The voxel centres as gto/gto_models.py:159-165 lays them out, from the resident grid's geometry."""
        lo, hi = self.workspace_bounds[:, 0], self.workspace_bounds[:, 1]
        m, r = self.field_margin, self.grid_resolution
        axes = [np.arange(lo[a] - m, hi[a] + m, r) for a in range(3)]
        return np.array(np.meshgrid(*axes, indexing="ij")).reshape((3, -1)).T

    def _depth_scene_for(self, dpc, epsilon, w_inside, grid_dpc=None):
        """Where the cost field of cloud `dpc` on the grid sized from cloud `grid_dpc` lives on the device: a
        depth_scene.Resident (handle, scene id, half, generation), or None if it cannot be resident.

        The resident scene is what the driver builds per object (examples/pybullet_gto_planning.py:176-190): field 0 =
        the cost field of the grid's cloud (all pixels of the depth image), field 1 = the cost field of a second cloud
        (the obstacle image with the target's mask) on the same grid, ONE gto_scene_from_depth call for both.  It is keyed by
        value on both clouds' inputs: a field is only ever served from the build it belongs to (its own image, mask and
        threshold; the grid it was asked on), as the half it is."""
        from .depth_scene import DEPTH_SCENE, CloudSnapshot, Resident, same_camera
        grid = grid_dpc if grid_dpc is not None else dpc
        if grid.target_mask is not None:
            return None  # (setup_points_field never leaves a masked cloud pending)
        h = self._util_handle()
        key = (float(self.grid_resolution), float(self.field_margin), float(epsilon), float(w_inside))
        c = self.__dict__.get("_depth_scene")
        cache_ok = c is not None and c["key"] == key and c["grid"].matches(grid) and c["gen"] == h.scene_generation(DEPTH_SCENE)
        if cache_ok and c["grid"].matches(dpc):
            return Resident(h, DEPTH_SCENE, 0, c["gen"])
        if cache_ok and c["obs"] is not None and c["obs"].matches(dpc):
            return Resident(h, DEPTH_SCENE, 1, c["gen"])
        # (A field of a PREVIOUS grid -- asked for before the last setup_points_field() -- still resolves to its own field on its
        # own grid: the resident scene is rebuilt for it, borrowers of the build that is gone share again, and the robot's
        # field_shape / origin go on describing the grid they were last set up for: a resident consumer reads the scene's own
        # geometry; one that takes the ARRAY of a stale field gets what the reference's driver gets when it keeps an array of
        # the previous object: an array that does not fit the grid, refused by gto_set_scene's size check.)
        is_grid_cloud = dpc is grid or CloudSnapshot(grid).matches(dpc)
        if not is_grid_cloud and not (same_camera(grid, dpc) and float(grid.threshold) == float(dpc.threshold)):
            return None  # another camera or cut-off than the grid's cloud: one call cannot serve both; the host path does
        obs = None if is_grid_cloud else dpc
        shape, origin, bounds = h.scene_from_depth(
            DEPTH_SCENE, grid.depth, grid.intrinsic_matrix, grid.camera_pose, None if obs is None else obs.target_mask, grid.threshold,
            self.grid_resolution, self.field_margin, epsilon, w_inside,
            depth_obstacle=None if obs is None or obs.depth is grid.depth else obs.depth)
        self._depth_scene = {"key": key, "grid": CloudSnapshot(grid), "obs": None if obs is None else CloudSnapshot(obs),
                             "gen": h.scene_generation(DEPTH_SCENE)}
        pend = self.__dict__.get("_pending_depth")
        if pend is not None and (pend is grid or self._depth_scene["grid"].matches(pend)):  # this IS the grid setup_points_field was asked for
            self.workspace_bounds = bounds
            self.origin = np.asarray(origin, dtype=np.float64).reshape((1, 3))
            self.field_shape = tuple(int(x) for x in shape)
            self.field_size = int(np.prod(shape))
        return Resident(h, DEPTH_SCENE, 0 if obs is None else 1, self._depth_scene["gen"])

    def field_geometry(self):
        if not hasattr(self, "field_shape"):
            raise RuntimeError("call setup_points_field() or setup_workspace_field() first")
        return tuple(int(s) for s in self.field_shape), self.origin.reshape(3).copy(), float(self.grid_resolution)

    def points_to_offsets_numpy(self, points):
        """gto/gto_models.py:190-201."""
        idx = (np.asarray(points, dtype=np.float64) - self.origin) / self.grid_resolution
        for a in range(3):
            idx[:, a] = np.clip(idx[:, a], 0, self.field_shape[a] - 1).astype(np.int32)
        off = idx[:, 2] + self.field_shape[2] * (idx[:, 1] + self.field_shape[1] * idx[:, 0])
        return np.clip(off, 0, self.field_size - 1).astype(np.int32)

    # ------------------------------------------------------------------ x-y occupancy grid (mobile base)
    def setup_occupancy_grid(self, points, epsilon=0.02):
        """gto/gto_models.py:218-244: grid nodes of the x-y plane that have an observed point (z > 0.01)
        within `epsilon`.  The reference asks a KD-tree for every node's nearest point; equivalently every
        point marks the nodes within `epsilon` of it (one scatter pass, no tree)."""
        points = np.asarray(points, dtype=np.float64)
        xys = points[points[:, 2] > 0.01, :2]
        m, r = self.field_margin, self.grid_resolution
        self.xlim_2d = [0, np.max(xys[:, 0])]
        self.ylim_2d = [np.min(xys[:, 1]), np.max(xys[:, 1])]
        self.occupancy_grid_origin = np.array([self.xlim_2d[0] - m, self.ylim_2d[0] - m]).reshape((1, 2))
        self.xgrid = np.arange(self.xlim_2d[0] - m, self.xlim_2d[1] + m, r)
        self.ygrid = np.arange(self.ylim_2d[0] - m, self.ylim_2d[1] + m, r)
        nx, ny = len(self.xgrid), len(self.ygrid)
        self.occupancy_grid_shape = (nx, ny)
        self.occupancy_grid_size = nx * ny
        grid = np.zeros((nx, ny))
        ix0 = np.rint((xys[:, 0] - self.xgrid[0]) / r).astype(np.int64)
        iy0 = np.rint((xys[:, 1] - self.ygrid[0]) / r).astype(np.int64)
        k = int(np.ceil(epsilon / r))
        for di in range(-k, k + 1):
            for dj in range(-k, k + 1):
                ix, iy = ix0 + di, iy0 + dj
                ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny)
                ix, iy, p = ix[ok], iy[ok], xys[ok]
                near = np.sqrt((p[:, 0] - self.xgrid[ix]) ** 2 + (p[:, 1] - self.ygrid[iy]) ** 2) < epsilon
                grid[ix[near], iy[near]] = 1.0
        self.occupancy_grid = grid.reshape(-1, 1)  # (size, 1), as the reference's KD-tree distances are

    def points_to_offsets_occupancy_numpy(self, points):
        """gto/gto_models.py:262-273."""
        xys = np.asarray(points, dtype=np.float64)[:, :2]
        idx = np.floor((xys - self.occupancy_grid_origin) / self.grid_resolution)
        for a in range(2):
            idx[:, a] = np.clip(idx[:, a], 0, self.occupancy_grid_shape[a] - 1).astype(np.int32)
        return (idx[:, 1] + self.occupancy_grid_shape[1] * idx[:, 0]).astype(np.int32)

    def compute_plan_cost(self, plan, sdf_cost_obstacle, base_position, handle=None):
        """gto/gto_models.py:204-215 for one plan (ndof, T) -> (cost, dist), evaluated on the GPU."""
        h = handle or self._util_handle()
        plan = np.asarray(plan, dtype=np.float64)
        if plan.shape[1] != h.T:
            raise NotImplementedError(f"plans must have T={h.T} waypoints on this handle")
        sid = self.SCRATCH_SCENE
        from .depth_scene import resident_of
        r = resident_of(sdf_cost_obstacle)
        if r is not None:  # resident on the device (depth_scene.py): the half this field is serves as the obstacle field
            if r.handle is h and r.half == 1:
                sid = r.sid
            else:
                h.share_scene(sid, r.handle, r.sid, all_from=r.half, obs_from=r.half)
        else:
            shape, origin, res = self.field_geometry()
            h.set_scene(sid, np.asarray(sdf_cost_obstacle), None, shape, origin, res, values_only=True)  # scored, never solved on
        cost, dist = h.plan_cost(sid, plan[None], base_position)
        return float(cost[0]), float(dist[0])

    def close(self):
        for h in self._handles.values():
            h.close()
        self._handles.clear()


def _rpy2r(rpy):
    """optas/spatialmath.py:186-211 ('zyx'), host-side, for rotating normals only."""
    r, p, y = rpy
    cr, sr, cp, sp_, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp_], [0, 1.0, 0], [-sp_, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx
