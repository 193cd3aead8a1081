"""ctypes binding of the C ABI in include/gto_solver.h (libgto_hip.so).

There is deliberately NO fallback: if the HIP library is missing or no GPU is present the calls
raise.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .robot_desc import RobotDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GTO_HIP_LIB", os.path.join(_HERE, "csrc", "libgto_hip.so"))  # override: A/B builds

GTO_MAX_FRAMES, GTO_MAX_LINKS, GTO_MAX_OPT, GTO_MAX_DOF = 32, 32, 16, 32
GRAD_CENTRAL_DIFF, GRAD_ZERO = 0, 1
STATUS_CONVERGED, STATUS_MAX_ITER, STATUS_NUMERICAL = 0, 1, 2

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pf = C.POINTER(C.c_float)


class CRobotDesc(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32), ("parent", _pi), ("joint_type", _pi), ("q_index", _pi),
        ("origin_xyz", _pd), ("origin_rpy", _pd), ("axis", _pd),
        ("ndof", C.c_int32), ("n_opt", C.c_int32), ("opt_index", _pi), ("lower", _pd), ("upper", _pd),
        ("n_links", C.c_int32), ("link_frame", _pi), ("visual_xyz", _pd), ("visual_rpy", _pd),
        ("n_points", C.c_int32), ("points", _pd), ("point_link", _pi),
        ("frame_ee", C.c_int32), ("frame_gripper", C.c_int32),
        ("n_gripper_points", C.c_int32), ("gripper_points", _pd),
    ]


class CSolverOpts(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("Tmax", C.c_double), ("standoff_offset", C.c_int32),
        ("w_obstacle", C.c_double), ("w_vel", C.c_double), ("max_iter", C.c_int32),
        ("tol_step", C.c_double), ("tol_rel_f", C.c_double), ("lambda0", C.c_double),
        ("grad_mode", C.c_int32),
    ]

    def copy(self) -> "CSolverOpts":
        o = CSolverOpts()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(CSolverOpts))
        return o


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: Optional[np.ndarray], typ):
    return None if a is None else a.ctypes.data_as(typ)


def pack_robot_desc(desc: RobotDesc, link_ee: str, link_gripper: str,
                    n_gripper_points: Optional[int] = None) -> Tuple[CRobotDesc, list]:
    """Build the C struct; the returned list keeps the backing arrays alive."""
    keep: List[np.ndarray] = []

    def k(a):
        keep.append(a)
        return a

    gp = desc.link_points(link_gripper)  # gto/gto_planner.py:37
    if n_gripper_points is not None:
        gp = gp[:n_gripper_points]
    opt = desc.opt_index
    c = CRobotDesc()
    c.n_frames = desc.n_frames
    c.parent = _p(k(_i32(desc.parent)), _pi)
    c.joint_type = _p(k(_i32(desc.joint_type)), _pi)
    c.q_index = _p(k(_i32(desc.q_index)), _pi)
    c.origin_xyz = _p(k(_f64(desc.origin_xyz)), _pd)
    c.origin_rpy = _p(k(_f64(desc.origin_rpy)), _pd)
    c.axis = _p(k(_f64(desc.axis)), _pd)
    c.ndof = desc.ndof
    c.n_opt = desc.n_opt
    c.opt_index = _p(k(_i32(opt)), _pi)
    c.lower = _p(k(_f64(desc.lower[opt])), _pd)
    c.upper = _p(k(_f64(desc.upper[opt])), _pd)
    c.n_links = desc.n_links
    c.link_frame = _p(k(_i32(desc.link_frame)), _pi)
    c.visual_xyz = _p(k(_f64(desc.visual_xyz)), _pd)
    c.visual_rpy = _p(k(_f64(desc.visual_rpy)), _pd)
    c.n_points = desc.n_points
    c.points = _p(k(_f64(desc.points)), _pd)
    c.point_link = _p(k(_i32(desc.point_link)), _pi)
    c.frame_ee = desc.frame_index(link_ee)
    c.frame_gripper = desc.frame_index(link_gripper)
    c.n_gripper_points = int(gp.shape[0])
    c.gripper_points = _p(k(_f64(gp)), _pd)
    return c, keep


# include/gto_solver.h
GTO_GRAD_CENTRAL_DIFF, GTO_GRAD_ZERO = 0, 1
GTO_STATUS_CONVERGED, GTO_STATUS_MAX_ITER, GTO_STATUS_NUMERICAL = 0, 1, 2
ABI_VERSION = 1007  # include/gto_solver.h GTO_ABI_VERSION (checked against gto_version() when the library is loaded)

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so with the SONAME of the system
    copy (libamdhip64.so.7) but link it by file name: if libgto_hip.so pulls in /opt/rocm's copy first and torch is
    imported afterwards, a second runtime is mapped next to it and finds no devices ("No HIP GPUs are available").
    Mapping torch's copy first (when a torch is installed) makes every later lookup, by SONAME or by file, land on it.
    Only done if the bundled copy has the SONAME of the runtime the library was linked against (same ABI; otherwise a
    warning and the system runtime); GTO_PRELOAD_TORCH_HIP=0 switches it off for processes that never import torch and
    want /opt/rocm's own copy."""
    import importlib.util
    if os.environ.get("GTO_PRELOAD_TORCH_HIP") == "0":
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        cand = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            if not _same_soname(cand, os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so")):
                import warnings
                warnings.warn(f"torch bundles a HIP runtime ({cand}) with another SONAME than /opt/rocm's: not preloading it; "
                              "import torch AFTER grasptrajopt_amd may then fail to find the GPU")
                return None
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return cand
    return None


def _soname(path):
    """DT_SONAME of an ELF shared object (None if it cannot be read)."""
    try:
        import re
        import subprocess
        out = subprocess.run(["readelf", "-d", os.path.realpath(path)], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"SONAME\)\s+Library soname: \[([^\]]+)\]", out)
        return m.group(1) if m else None
    except Exception:
        return None


def _same_soname(a, b):
    sa, sb = _soname(a), _soname(b)
    return sa is None or sb is None or sa == sb  # unreadable: do as before


def load_library(path: Optional[str] = None):
    """Load libgto_hip.so (built by __graft_entry__.build()). Raises if it is not there."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"HIP library not found at {p}: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the GTO solve path.")
    _preload_hip_runtime()
    lib = C.CDLL(p)
    lib.gto_version.restype = C.c_int32
    # one ABI number for wrapper and library: a library named by GTO_HIP_LIB for an A/B run has to be a build of THIS
    # interface -- an older one that lacks a symbol, or has it with other arguments (gto_scene_from_depth once got a pointer
    # in the middle of its signature), is refused here instead of being called with shifted arguments
    v = int(lib.gto_version())
    if v != ABI_VERSION:
        raise RuntimeError(f"{p}: ABI version {v}, this wrapper speaks {ABI_VERSION} (include/gto_solver.h GTO_ABI_VERSION): "
                           "rebuild the library from this tree (__graft_entry__.build())")
    H = C.c_void_p
    lib.gto_default_opts.argtypes = [C.POINTER(CSolverOpts)]
    lib.gto_default_opts.restype = None
    lib.gto_create.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CSolverOpts), C.c_int, C.POINTER(H)]
    lib.gto_destroy.argtypes = [H]
    lib.gto_destroy.restype = None
    lib.gto_last_error.argtypes = [H]
    lib.gto_last_error.restype = C.c_char_p
    lib.gto_set_opts.argtypes = [H, C.POINTER(CSolverOpts)]
    lib.gto_set_scene.argtypes = [H, C.c_int32, _pf, _pf, _pi, _pd, C.c_double]
    lib.gto_set_scene_values.argtypes = [H, C.c_int32, _pf, _pf, _pi, _pd, C.c_double]
    lib.gto_drop_scene.argtypes = [H, C.c_int32]
    solve_args = [H, C.c_int32, C.c_int32] + [C.c_void_p] * 12
    lib.gto_solve_batch.argtypes = solve_args
    lib.gto_solve_batch_device.argtypes = solve_args + [C.c_void_p]
    lib.gto_last_kernel_time.argtypes = [H, _pd, _pi]
    lib.gto_set_profiling.argtypes = [H, C.c_int32]
    lib.gto_last_kernel_work.argtypes = [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gto_last_kernel_profile.argtypes = [H, C.c_int32, _pd, _pi, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gto_last_kernel_profile.restype = C.c_int
    lib.gto_set_stream.argtypes = [H, C.c_void_p]
    lib.gto_set_mode.argtypes = [H, C.c_int32]
    lib.gto_set_lanes.argtypes = [H, C.c_int32, C.c_int32, C.c_int32]
    lib.gto_set_lane_streams.argtypes = [H, C.c_int32, C.POINTER(C.c_void_p)]
    lib.gto_share_scene.argtypes = [H, C.c_int32, H, C.c_int32]
    lib.gto_share_scene_halves.argtypes = [H, C.c_int32, H, C.c_int32, C.c_int32, C.c_int32]
    lib.gto_share_scene_halves.restype = C.c_int
    lib.gto_scene_from_depth.argtypes = [H, C.c_int32, _pf, C.c_int32, C.c_int32, _pd, _pd, _pd, _pd, C.POINTER(C.c_uint8), _pf, C.c_double,
                                         C.c_double, C.c_double, C.c_float, C.c_float, _pi, _pd, _pd]
    lib.gto_scene_from_depth.restype = C.c_int
    lib.gto_get_scene_fields.argtypes = [H, C.c_int32, _pf, _pf]
    lib.gto_get_scene_fields.restype = C.c_int
    lib.gto_eval_fk.argtypes = [H, C.c_int32, _pd, _pd]
    lib.gto_eval_points.argtypes = [H, C.c_int32, C.c_int32, _pd, _pd, C.c_int32, _pd, _pi, _pd, _pd]
    lib.gto_eval_points_hessian.argtypes = [H, C.c_int32, C.c_int32, _pd, _pd, C.c_int32, _pd]
    lib.gto_eval_points_hessian.restype = C.c_int
    lib.gto_eval_objective.argtypes = [H, C.c_int32, C.c_int32, _pi, _pd, _pi, _pd, _pd, _pd, _pd, _pd, _pd, _pi]
    lib.gto_eval_obstacle_normal_eq.argtypes = [H, C.c_int32, _pi, _pd, _pd, _pd, _pd, _pd]
    lib.gto_plan_cost.argtypes = [H, C.c_int32, C.c_int32, _pd, _pd, _pd, _pd]
    lib.gto_solve_ik_batch.argtypes = [H, C.c_int32, _pi, _pd, _pd, _pd, C.c_int32, _pd, _pd, _pi, _pi]
    lib.gto_solve_base_batch.argtypes = [H, C.c_int32, C.c_int32, _pi, _pd, _pd, C.c_double, C.c_int32, _pd, _pd, _pd, _pi, _pi]
    lib.gto_eval_base_objective.argtypes = [H, C.c_int32, C.c_int32, _pi, _pd, _pd, _pd, C.c_double, _pd]
    _pu8 = C.POINTER(C.c_uint8)
    lib.gto_depth_sdf_cost.argtypes = [C.c_int, _pf, C.c_int32, C.c_int32, _pd, _pd, _pd, _pd, _pu8, C.c_double, _pd, C.c_int64,
                                       C.c_float, C.c_float, _pf, _pu8, _pf, _pd, _pu8]
    for fn in ("gto_create", "gto_set_opts", "gto_set_scene", "gto_set_scene_values", "gto_drop_scene", "gto_solve_batch",
               "gto_solve_batch_device", "gto_last_kernel_time", "gto_last_kernel_work", "gto_set_profiling", "gto_set_stream", "gto_set_mode", "gto_set_lanes", "gto_set_lane_streams", "gto_share_scene",
               "gto_eval_fk", "gto_eval_points", "gto_eval_points_hessian", "gto_eval_objective", "gto_eval_obstacle_normal_eq", "gto_plan_cost", "gto_solve_ik_batch",
               "gto_solve_base_batch", "gto_eval_base_objective", "gto_depth_sdf_cost"):
        getattr(lib, fn).restype = C.c_int
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "gto_default_opts", "gto_version", "gto_create", "gto_destroy", "gto_last_error", "gto_set_opts",
    "gto_set_scene", "gto_set_scene_values", "gto_drop_scene", "gto_solve_batch", "gto_solve_batch_device",
    "gto_last_kernel_time", "gto_last_kernel_work", "gto_last_kernel_profile", "gto_set_profiling", "gto_set_stream", "gto_set_mode", "gto_set_lanes", "gto_set_lane_streams", "gto_share_scene", "gto_share_scene_halves",
    "gto_eval_fk", "gto_eval_points", "gto_eval_points_hessian",
    "gto_eval_objective", "gto_eval_obstacle_normal_eq", "gto_plan_cost", "gto_solve_ik_batch", "gto_solve_base_batch",
    "gto_eval_base_objective", "gto_depth_sdf_cost", "gto_scene_from_depth", "gto_get_scene_fields",
)


def default_opts() -> CSolverOpts:
    o = CSolverOpts()
    load_library().gto_default_opts(C.byref(o))
    return o


class GTOError(RuntimeError):
    pass


class SolverHandle:
    """Owns one gto_handle (one HIP device, one stream)."""

    def __init__(self, desc: RobotDesc, link_ee: str, link_gripper: str,
                 opts: Optional[CSolverOpts] = None, device: int = -1,
                 n_gripper_points: Optional[int] = None):
        self.lib = load_library()
        self.desc = desc
        self.opts = opts.copy() if opts is not None else default_opts()
        self._cdesc, self._keep = pack_robot_desc(desc, link_ee, link_gripper, n_gripper_points)
        h = C.c_void_p()
        rc = self.lib.gto_create(C.byref(self._cdesc), C.byref(self.opts), device, C.byref(h))
        if rc != 0:
            msg = self.lib.gto_last_error(None)
            raise GTOError(f"gto_create failed ({rc}): {msg.decode() if msg else ''}")
        self._h = h
        self.scenes = {}
        self._scene_gen = {}  # scene id -> number of times it was written (set_scene, scene_from_depth, share_scene, drop_scene)

    # -------------------------------------------------------------- helpers
    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.gto_last_error(self._h)
            raise GTOError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gto_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def T(self) -> int:
        return int(self.opts.T)

    def set_opts(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.opts, k):
                raise AttributeError(k)
            setattr(self.opts, k, v)
        self._check(self.lib.gto_set_opts(self._h, C.byref(self.opts)), "gto_set_opts")

    def set_scene(self, scene_id: int, c_all, c_obs, shape: Sequence[int], origin, res: float, values_only: bool = False):
        """values_only: gto_set_scene_values, the fields without the solver's records and distance fields (such a scene
        serves plan_cost and eval_points only)."""
        ca = np.ascontiguousarray(c_all, dtype=np.float32).reshape(-1)
        co = None if c_obs is None else np.ascontiguousarray(c_obs, dtype=np.float32).reshape(-1)
        shp = _i32(list(shape))
        n = int(shp[0]) * int(shp[1]) * int(shp[2])
        if ca.size != n or (co is not None and co.size != n):
            raise ValueError(f"field size {ca.size} does not match shape {tuple(shp)}")
        org = _f64(np.asarray(origin).reshape(3))
        fn = self.lib.gto_set_scene_values if values_only else self.lib.gto_set_scene
        self._check(fn(self._h, scene_id, _p(ca, _pf), _p(co, _pf), _p(shp, _pi), _p(org, _pd), float(res)),
                    "gto_set_scene_values" if values_only else "gto_set_scene")
        self.scenes[scene_id] = (tuple(int(s) for s in shp), org.copy(), float(res))
        self._bump(scene_id)

    def _bump(self, scene_id: int):
        """A scene of this handle was written: handles that borrowed it (share_scene) hold pointers into buffers that are
        gone.  Borrowers remember the generation they shared and share again when it has moved (scene_generation)."""
        self._scene_gen[int(scene_id)] = self._scene_gen.get(int(scene_id), 0) + 1

    def scene_generation(self, scene_id: int) -> int:
        return self._scene_gen.get(int(scene_id), 0)

    def scene_from_depth(self, scene_id: int, depth, K, cam_pose, target_mask=None, threshold=1.5, grid_res=0.05, margin=0.4,
                         epsilon=0.02, w_inside=1.0, depth_obstacle=None):
        """gto_scene_from_depth: depth image -> both cost fields resident as scene `scene_id` (voxel records and distance
        fields included); returns (shape, origin, bounds (3, 2)) of the grid, nothing else leaves the device.
        depth_obstacle: the image of the second cloud (the driver's copy with the target's pixels at the threshold,
        examples/pybullet_gto_planning.py:187-189); None: the same image."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        Hh, Ww = depth.shape
        dobs = None if depth_obstacle is None else np.ascontiguousarray(depth_obstacle, dtype=np.float32)
        if dobs is not None and dobs.shape != depth.shape:
            raise ValueError("depth_obstacle must have the shape of depth")
        K = _f64(K).reshape(3, 3)
        cam = _f64(cam_pose).reshape(4, 4)
        Kinv, cinv = np.ascontiguousarray(np.linalg.inv(K)), np.ascontiguousarray(np.linalg.inv(cam))
        mask = None if target_mask is None else np.ascontiguousarray(target_mask, dtype=np.uint8)
        shp, org, bnd = np.zeros(3, np.int32), np.zeros(3), np.zeros(6)
        pu8 = C.POINTER(C.c_uint8)
        self._check(self.lib.gto_scene_from_depth(self._h, scene_id, _p(depth, _pf), Hh, Ww, _p(K, _pd), _p(Kinv, _pd), _p(cam, _pd), _p(cinv, _pd),
                                                  None if mask is None else mask.ctypes.data_as(pu8), _p(dobs, _pf), float(threshold), float(grid_res), float(margin),
                                                  float(epsilon), float(w_inside), _p(shp, _pi), _p(org, _pd), _p(bnd, _pd)), "gto_scene_from_depth")
        self.scenes[scene_id] = (tuple(int(x) for x in shp), org.copy(), float(grid_res))
        self._bump(scene_id)
        return tuple(int(x) for x in shp), org, np.stack((bnd[:3], bnd[3:]), axis=1)

    def scene_fields(self, scene_id: int):
        """(c_all, c_obs) of a resident scene as float32 arrays (device to host)."""
        shape = self.scenes[scene_id][0]
        n = int(np.prod(shape))
        ca, co = np.empty(n, np.float32), np.empty(n, np.float32)
        self._check(self.lib.gto_get_scene_fields(self._h, scene_id, _p(ca, _pf), _p(co, _pf)), "gto_get_scene_fields")
        return ca, co

    def drop_scene(self, scene_id: int):
        self._check(self.lib.gto_drop_scene(self._h, scene_id), "gto_drop_scene")
        self.scenes.pop(scene_id, None)
        self._bump(scene_id)

    # -------------------------------------------------------------- solve
    def solve_batch(self, scene_id, qc, goals, n_goals, standoff, base_pos, Q0, out=None):
        """gto_solve_batch through host arrays.  ``out``: optional tuple (Q, dQ, cost, iters, status) of C-contiguous arrays
        to write into (shapes (B,ndof,T) f64, (B,ndof,T-1) f64, (B,) f64, (B,) i32, (B,) i32): a caller that solves batch
        after batch keeps its result arrays instead of having fresh pages mapped and faulted in on every call."""
        d, T = self.desc, self.T
        qc = _f64(qc).reshape(-1, d.ndof)
        B = qc.shape[0]
        if B == 0:  # empty batch: nothing to solve
            return (np.empty((0, d.ndof, T)), np.empty((0, d.ndof, T - 1)), np.empty(0),
                    np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32))
        goals = _f64(goals).reshape(B, -1, 16)
        n_max = goals.shape[1]
        n_goals = _i32(np.broadcast_to(np.asarray(n_goals), (B,)))
        scene_id = _i32(np.broadcast_to(np.asarray(scene_id), (B,)))
        so = None if standoff is None else _f64(np.broadcast_to(_f64(standoff).reshape(-1, 16), (B, 16)))
        base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (B, 3)))
        Q0 = _f64(Q0).reshape(B, d.ndof, T)
        if out is not None:
            Q, dQ, cost, iters, status = out
            want = (((B, d.ndof, T), np.float64), ((B, d.ndof, T - 1), np.float64), ((B,), np.float64), ((B,), np.int32), ((B,), np.int32))
            for a, (shp, dt) in zip(out, want):
                if a.shape != shp or a.dtype != dt or not a.flags.c_contiguous:
                    raise ValueError(f"solve_batch(out=): expected a C-contiguous {np.dtype(dt).name} array of shape {shp}, got {a.dtype} {a.shape}")
        else:
            Q = np.empty((B, d.ndof, T))
            dQ = np.empty((B, d.ndof, T - 1))
            cost = np.empty(B)
            iters = np.empty(B, dtype=np.int32)
            status = np.empty(B, dtype=np.int32)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        rc = self.lib.gto_solve_batch(self._h, B, n_max, vp(scene_id), vp(qc), vp(goals), vp(n_goals),
                                      vp(so), vp(base), vp(Q0), vp(Q), vp(dQ), vp(cost), vp(iters), vp(status))
        self._check(rc, "gto_solve_batch")
        return Q, dQ, cost, iters, status

    def solve_batch_device(self, B, n_max, scene_id, qc, goals, n_goals, standoff, base_pos, Q0,
                           Q_out, dQ_out, cost_out, iters_out, status_out, stream=None):
        """All arguments are device pointers (ints, e.g. torch.Tensor.data_ptr()) or None."""
        vp = lambda a: None if a is None else C.c_void_p(int(a))
        rc = self.lib.gto_solve_batch_device(self._h, B, n_max, vp(scene_id), vp(qc), vp(goals), vp(n_goals),
                                             vp(standoff), vp(base_pos), vp(Q0), vp(Q_out), vp(dQ_out),
                                             vp(cost_out), vp(iters_out), vp(status_out), vp(stream))
        self._check(rc, "gto_solve_batch_device")

    def share_scene(self, scene_id, src: "SolverHandle", src_scene_id=None, all_from: int = 0, obs_from: int = 1):
        """Use a scene that lives in another handle on the same GPU without a second copy.  all_from / obs_from: which of
        the source's two fields (0: sdf_cost_all, 1: sdf_cost_obstacle) this handle sees as its sdf_cost_all / sdf_cost_obstacle
        (gto_share_scene_halves)."""
        src_id = int(scene_id if src_scene_id is None else src_scene_id)
        if (all_from, obs_from) == (0, 1):
            self._check(self.lib.gto_share_scene(self._h, int(scene_id), src._h, src_id), "gto_share_scene")
        else:
            self._check(self.lib.gto_share_scene_halves(self._h, int(scene_id), src._h, src_id, int(all_from), int(obs_from)), "gto_share_scene_halves")
        if src_id in src.scenes:
            self.scenes[int(scene_id)] = src.scenes[src_id]
        self._bump(scene_id)

    def set_stream(self, stream):
        """Bind every launch/copy of this handle to the caller's HIP stream (an int such as
        torch.cuda.Stream.cuda_stream); None restores a private stream."""
        self._check(self.lib.gto_set_stream(self._h, None if stream is None else C.c_void_p(int(stream))), "gto_set_stream")

    MODE_ROUNDS, MODE_SINGLE_LAUNCH = 0, 1

    def set_mode(self, mode: int):
        """MODE_ROUNDS (default): evaluate / step launches over slots; MODE_SINGLE_LAUNCH: one launch per call."""
        self._check(self.lib.gto_set_mode(self._h, int(mode)), "gto_set_mode")

    def set_lanes(self, max_lanes: int = 1, min_per_lane: int = 256, adopt_below: int = 0):
        """Lanes of a solve call (include/gto_solver.h, gto_set_lanes): streams the call's instances are dealt to, one host
        thread of the call each.  The defaults are the library's (one lane: no thread but the caller's): several lanes are
        asked for explicitly, e.g. set_lanes(4, 256, 0)."""
        self._check(self.lib.gto_set_lanes(self._h, int(max_lanes), int(min_per_lane), int(adopt_below)), "gto_set_lanes")

    def set_lane_streams(self, streams=()):
        """The lanes' HIP streams (raw handles, e.g. torch.cuda.Stream(...).cuda_stream); () = the handle's own."""
        arr = (C.c_void_p * max(1, len(streams)))(*[C.c_void_p(int(x)) for x in streams])
        self._check(self.lib.gto_set_lane_streams(self._h, len(streams), arr), "gto_set_lane_streams")

    def set_profiling(self, enabled: bool):
        self._check(self.lib.gto_set_profiling(self._h, int(enabled)), "gto_set_profiling")

    def last_kernel_time(self):
        ms = C.c_double()
        n = C.c_int32()
        self._check(self.lib.gto_last_kernel_time(self._h, C.byref(ms), C.byref(n)), "gto_last_kernel_time")
        return ms.value, n.value

    def last_kernel_work(self):
        """(surface points gathered, chunk spheres tested) by the dominant kernel during the last profiled solve."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.gto_last_kernel_work(self._h, C.byref(a), C.byref(b)), "gto_last_kernel_work")
        return a.value, b.value

    PROF_VARIANTS = ("k_obstacle_gram<8,1>", "k_obstacle_gram<8,8>", "k_lm_step<4,1>", "k_lm_step<8,4>")

    def last_kernel_profile(self):
        """Per kernel variant of the last profiled solve: {name: (ms, launches, workgroups, points gathered)}."""
        out = {}
        names = self.PROF_VARIANTS
        if self.desc.n_opt > 8:  # the kernels of the robots with nine to sixteen optimised joints (one step kernel, no few-instance variants)
            names = ("k_obstacle_gram<16,1>", "k_obstacle_gram<16,8>", "k_lm_step_wide<16>", "k_lm_step<8,4>")
        for v, name in enumerate(names):
            ms, n, wg, pts = C.c_double(), C.c_int32(), C.c_uint64(), C.c_uint64()
            self._check(self.lib.gto_last_kernel_profile(self._h, v, C.byref(ms), C.byref(n), C.byref(wg), C.byref(pts)), "gto_last_kernel_profile")
            out[name] = (ms.value, n.value, wg.value, pts.value)
        return out

    # -------------------------------------------------------------- evaluation entry points
    def eval_fk(self, q):
        q = _f64(q).reshape(-1, self.desc.ndof)
        out = np.empty((q.shape[0], self.desc.n_frames, 4, 4))
        self._check(self.lib.gto_eval_fk(self._h, q.shape[0], _p(q, _pd), _p(out, _pd)), "gto_eval_fk")
        return out

    def eval_points(self, scene_id, q, base_pos, use_obs=False, want_field=True, want=None):
        """World surface points, voxel offsets, field values and field gradients of configurations q (nq, ndof).
        ``want``: the outputs to bring back, any of "xyz", "off", "val", "grad" (default: all, or xyz alone with
        want_field=False); the others are None and cost neither device memory nor a transfer."""
        q = _f64(q).reshape(-1, self.desc.ndof)
        nq, P = q.shape[0], self.desc.n_points
        base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (nq, 3)))
        if want is None:
            want = ("xyz", "off", "val", "grad") if want_field else ("xyz",)
        xyz = np.empty((nq, P, 3)) if "xyz" in want else None
        off = np.empty((nq, P), dtype=np.int32) if "off" in want else None
        val = np.empty((nq, P)) if "val" in want else None
        grad = np.empty((nq, P, 3)) if "grad" in want else None
        self._check(self.lib.gto_eval_points(self._h, scene_id, nq, _p(q, _pd), _p(base, _pd), int(use_obs),
                                             _p(xyz, _pd), _p(off, _pi), _p(val, _pd), _p(grad, _pd)),
                    "gto_eval_points")
        return xyz, off, val, grad

    def eval_points_hessian(self, scene_id, q, base_pos, use_obs=False):
        """Hessian of the selected cost field at the surface points of configurations q: (nq, P, 3, 3), gto/sdf_callback.py:165-183."""
        q = _f64(q).reshape(-1, self.desc.ndof)
        nq, P = q.shape[0], self.desc.n_points
        base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (nq, 3)))
        out = np.empty((nq, P, 3, 3))
        self._check(self.lib.gto_eval_points_hessian(self._h, scene_id, nq, _p(q, _pd), _p(base, _pd), int(use_obs), _p(out, _pd)), "gto_eval_points_hessian")
        return out

    def eval_objective(self, scene_id, goals, n_goals, standoff, base_pos, Q):
        d, T = self.desc, self.T
        Q = _f64(Q).reshape(-1, d.ndof, T)
        B = Q.shape[0]
        goals = _f64(goals).reshape(B, -1, 16)
        n_max = goals.shape[1]
        n_goals = _i32(np.broadcast_to(np.asarray(n_goals), (B,)))
        scene_id = _i32(np.broadcast_to(np.asarray(scene_id), (B,)))
        so = None if standoff is None else _f64(np.broadcast_to(_f64(standoff).reshape(-1, 16), (B, 16)))
        base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (B, 3)))
        fg, fo, fv = np.empty(B), np.empty(B), np.empty(B)
        am = np.empty(B, dtype=np.int32)
        self._check(self.lib.gto_eval_objective(self._h, B, n_max, _p(scene_id, _pi), _p(goals, _pd),
                                                _p(n_goals, _pi), _p(so, _pd), _p(base, _pd), _p(Q, _pd),
                                                _p(fg, _pd), _p(fo, _pd), _p(fv, _pd), _p(am, _pi)),
                    "gto_eval_objective")
        return fg, fo, fv, am

    def eval_obstacle_normal_eq(self, scene_id, base_pos, Q):
        d, T, n = self.desc, self.T, self.desc.n_opt
        Q = _f64(Q).reshape(-1, d.ndof, T)
        B = Q.shape[0]
        scene_id = _i32(np.broadcast_to(np.asarray(scene_id), (B,)))
        base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (B, 3)))
        JtJ, Jtr, ss = np.empty((B, T, n, n)), np.empty((B, T, n)), np.empty((B, T))
        self._check(self.lib.gto_eval_obstacle_normal_eq(self._h, B, _p(scene_id, _pi), _p(base, _pd),
                                                         _p(Q, _pd), _p(JtJ, _pd), _p(Jtr, _pd), _p(ss, _pd)),
                    "gto_eval_obstacle_normal_eq")
        return JtJ, Jtr, ss

    def solve_ik_batch(self, scene_id, q0, goals, base_pos=None, max_iter=50):
        """IK for B goal poses (gto/ik_solver.py:78-110); scene_id None = no collision term.
        Returns (q (B,ndof), cost (B,), iters (B,), status (B,))."""
        d = self.desc
        q0 = _f64(q0).reshape(-1, d.ndof)
        B = q0.shape[0]
        goals = _f64(goals).reshape(B, 16)
        sid = None if scene_id is None else _i32(np.broadcast_to(np.asarray(scene_id), (B,)))
        base = None if base_pos is None else _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (B, 3)))
        q, cost = np.empty((B, d.ndof)), np.empty(B)
        iters, status = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        if B:
            self._check(self.lib.gto_solve_ik_batch(self._h, B, _p(sid, _pi), _p(q0, _pd), _p(goals, _pd), _p(base, _pd),
                                                    int(max_iter), _p(q, _pd), _p(cost, _pd), _p(iters, _pi), _p(status, _pi)),
                        "gto_solve_ik_batch")
        return q, cost, iters, status

    def solve_base_batch(self, qc, goals, n_goals=None, effort_weight=0.01, max_iter=100):
        """Base placement for B goal sets (gto/base_planner.py:35-123): qc (B,ndof), goals (B,n_max,4,4).
        Returns (y (B,3) = x, y, theta, q (B,n_max,ndof), cost (B,), iters (B,), status (B,))."""
        d = self.desc
        qc = _f64(qc).reshape(-1, d.ndof)
        B = qc.shape[0]
        goals = _f64(goals).reshape(B, -1, 16)
        n_max = goals.shape[1]
        ng = _i32(np.broadcast_to(np.asarray(n_max if n_goals is None else n_goals), (B,)))
        y, q, cost = np.empty((B, 3)), np.empty((B, n_max, d.ndof)), np.empty(B)
        iters, status = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        if B:
            self._check(self.lib.gto_solve_base_batch(self._h, B, n_max, _p(ng, _pi), _p(qc, _pd), _p(goals, _pd),
                                                      float(effort_weight), int(max_iter), _p(y, _pd), _p(q, _pd),
                                                      _p(cost, _pd), _p(iters, _pi), _p(status, _pi)),
                        "gto_solve_base_batch")
        return y, q, cost, iters, status

    def eval_base_objective(self, y, q, goals, n_goals=None, effort_weight=0.01):
        """Base-placement objective (gto/base_planner.py:57-87) at y (B,3), q (B,n_max,ndof), goals (B,n_max,4,4)."""
        d = self.desc
        y = _f64(y).reshape(-1, 3)
        B = y.shape[0]
        goals = _f64(goals).reshape(B, -1, 16)
        n_max = goals.shape[1]
        q = _f64(q).reshape(B, n_max, d.ndof)
        ng = _i32(np.broadcast_to(np.asarray(n_max if n_goals is None else n_goals), (B,)))
        cost = np.empty(B)
        if B:
            self._check(self.lib.gto_eval_base_objective(self._h, B, n_max, _p(ng, _pi), _p(y, _pd), _p(q, _pd), _p(goals, _pd),
                                                         float(effort_weight), _p(cost, _pd)), "gto_eval_base_objective")
        return cost

    def plan_cost(self, scene_id, plans, base_pos):
        d, T = self.desc, self.T
        plans = _f64(plans).reshape(-1, d.ndof, T)
        n = plans.shape[0]
        base = _f64(np.asarray(base_pos).reshape(3))
        cost, dist = np.empty(n), np.empty(n)
        self._check(self.lib.gto_plan_cost(self._h, scene_id, n, _p(plans, _pd), _p(base, _pd),
                                           _p(cost, _pd), _p(dist, _pd)), "gto_plan_cost")
        return cost, dist
