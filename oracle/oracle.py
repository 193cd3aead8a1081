"""ctypes wrapper of the CPU oracle (oracle/gto_oracle.c -> libgto_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

from grasptrajopt_amd._capi import CRobotDesc, CSolverOpts, pack_robot_desc
from grasptrajopt_amd.robot_desc import RobotDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgto_oracle.so")

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pf = C.POINTER(C.c_float)


class CScene(C.Structure):
    _fields_ = [("c_all", _pf), ("c_obs", _pf), ("shape", C.c_int32 * 3), ("origin", C.c_double * 3),
                ("res", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gto_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "gto_solver.h")
    if (force or not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgto_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def reference_opts(**kw) -> CSolverOpts:
    """The reference's planner constants (gto/gto_planner.py:25-30,131,135,141) + solver defaults.
    Must equal gto_default_opts() of the HIP library (tests/test_capi_cpu.py checks it)."""
    o = CSolverOpts()
    o.T, o.Tmax, o.standoff_offset = 50, 10.0, -10
    o.w_obstacle, o.w_vel = 10.0, 0.01
    o.max_iter = 100
    o.tol_step, o.tol_rel_f, o.lambda0 = 1e-7, 1e-8, 1e-3
    o.grad_mode = 0
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


# ---------------------------------------------------------------------- stand-alone numerics
def rpy2r(rpy):
    R = np.empty(9)
    lib().orc_rpy2r(_p(_f64(rpy), _pd), _p(R, _pd))
    return R.reshape(3, 3)


def angvec2r(theta, axis):
    R = np.empty(9)
    lib().orc_angvec2r.argtypes = [C.c_double, _pd, _pd]
    lib().orc_angvec2r(float(theta), _p(_f64(axis), _pd), _p(R, _pd))
    return R.reshape(3, 3)


def points_to_offsets(xyz, origin, res, shape):
    xyz = _f64(xyz).reshape(-1, 3)
    off = np.empty(xyz.shape[0], dtype=np.int32)
    f = lib().orc_points_to_offsets
    f.argtypes = [_pd, C.c_int, _pd, C.c_double, _pi, _pi]
    f(_p(xyz, _pd), xyz.shape[0], _p(_f64(origin).reshape(3), _pd), float(res), _p(_i32(shape), _pi), _p(off, _pi))
    return off


def sdf_eval(data, shape, origin, res, xyz, want_hess=True):
    data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
    xyz = _f64(xyz).reshape(-1, 3)
    n = xyz.shape[0]
    val, jac = np.empty(n), np.empty((n, 3))
    hes = np.empty((n, 3, 3)) if want_hess else None
    f = lib().orc_sdf_eval
    f.argtypes = [_pf, _pi, _pd, C.c_double, C.c_int, _pd, _pd, _pd, _pd]
    f(_p(data, _pf), _p(_i32(shape), _pi), _p(_f64(origin).reshape(3), _pd), float(res), n, _p(xyz, _pd),
      _p(val, _pd), _p(jac, _pd), _p(hes, _pd))
    return val, jac, hes


def sdf_cost_map(signed_dist, inside, epsilon=0.02, w_inside=1.0):
    d = np.ascontiguousarray(signed_dist, dtype=np.float32).reshape(-1)
    ins = np.ascontiguousarray(inside, dtype=np.uint8).reshape(-1)
    out = np.empty_like(d)
    f = lib().orc_sdf_cost_map
    f.argtypes = [C.c_int, _pf, C.POINTER(C.c_uint8), C.c_float, C.c_float, _pf]
    f(d.size, _p(d, _pf), ins.ctypes.data_as(C.POINTER(C.c_uint8)), epsilon, w_inside, _p(out, _pf))
    return out


def depth_sdf_cost(depth, K, cam, target_mask, threshold, query, epsilon=0.02, w_inside=1.0):
    """DepthPointCloud(depth, K, cam, target_mask, threshold) -> points, then get_sdf / is_outside /
    get_sdf_cost at `query` (mesh_to_sdf/depth_point_cloud.py:9-141).  Returns (points (N,3) in pixel
    order, sdf f32 (nq,), inside bool (nq,), cost f32 (nq,))."""
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    H, W = depth.shape
    K = _f64(K).reshape(3, 3)
    cam = _f64(cam).reshape(4, 4)
    Kinv, cam_inv = _f64(np.linalg.inv(K)), _f64(np.linalg.inv(cam))
    tm = None if target_mask is None else np.ascontiguousarray(target_mask, dtype=np.uint8).reshape(H, W)
    pts = np.empty((H * W, 3))
    valid = np.empty(H * W, dtype=np.uint8)
    pu8 = C.POINTER(C.c_uint8)
    f = lib().orc_depth_backproject
    f.argtypes = [_pf, C.c_int, C.c_int, _pd, _pd, pu8, C.c_double, _pd, pu8]
    f.restype = C.c_int
    f(_p(depth, _pf), H, W, _p(Kinv, _pd), _p(cam, _pd), None if tm is None else tm.ctypes.data_as(pu8), float(threshold),
      _p(pts, _pd), valid.ctypes.data_as(pu8))
    query = _f64(query).reshape(-1, 3)
    nq = query.shape[0]
    sdf = np.empty(nq, dtype=np.float32)
    inside = np.empty(nq, dtype=np.uint8)
    g = lib().orc_depth_sdf
    g.argtypes = [_pd, pu8, C.c_int, C.c_int, _pf, _pd, _pd, _pd, C.c_long, _pf, pu8]
    g.restype = None
    g(_p(pts, _pd), valid.ctypes.data_as(pu8), H, W, _p(depth, _pf), _p(K, _pd), _p(cam_inv, _pd), _p(query, _pd), nq,
      _p(sdf, _pf), inside.ctypes.data_as(pu8))
    cost = sdf_cost_map(sdf, inside, epsilon, w_inside)
    return pts[valid.astype(bool)], sdf, inside.astype(bool), cost


def interpolate_waypoints(waypoints, n, m):
    w = _f64(waypoints).reshape(2, m)
    out = np.empty((n, m))
    f = lib().orc_interpolate_waypoints
    f.argtypes = [_pd, C.c_int, C.c_int, _pd]
    f(_p(w, _pd), n, m, _p(out, _pd))
    return out


# ---------------------------------------------------------------------- robot-bound oracle
class Oracle:
    """Same surface as grasptrajopt_amd._capi.SolverHandle, computed on the CPU in FP64."""

    def __init__(self, desc: RobotDesc, link_ee: str, link_gripper: str,
                 opts: Optional[CSolverOpts] = None, n_gripper_points: Optional[int] = None):
        self.lib = lib()
        self.desc = desc
        self.opts = opts.copy() if opts is not None else reference_opts()
        self._cdesc, self._keep = pack_robot_desc(desc, link_ee, link_gripper, n_gripper_points)
        self._scenes = {}
        self._scene_keep = {}

    @property
    def T(self):
        return int(self.opts.T)

    def set_opts(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.opts, k):
                raise AttributeError(k)
            setattr(self.opts, k, v)

    def set_scene(self, scene_id, c_all, c_obs, shape, origin, res):
        ca = np.ascontiguousarray(c_all, dtype=np.float32).reshape(-1)
        co = ca if c_obs is None else np.ascontiguousarray(c_obs, dtype=np.float32).reshape(-1)
        s = CScene()
        s.c_all = _p(ca, _pf)
        s.c_obs = _p(co, _pf)
        for i in range(3):
            s.shape[i] = int(shape[i])
            s.origin[i] = float(np.asarray(origin).reshape(3)[i])
        s.res = float(res)
        self._scenes[scene_id] = s
        self._scene_keep[scene_id] = (ca, co)

    def _scene_array(self):
        n = max(self._scenes) + 1 if self._scenes else 1
        arr = (CScene * n)()
        for k, s in self._scenes.items():
            arr[k] = s
        return arr

    def _null_scene(self):
        s = CScene()
        for i in range(3):
            s.shape[i] = 1
        s.res = 1.0
        return s

    def eval_fk(self, q):
        q = _f64(q).reshape(-1, self.desc.ndof)
        out = np.empty((q.shape[0], self.desc.n_frames, 4, 4))
        f = self.lib.orc_eval_fk
        f.argtypes = [C.POINTER(CRobotDesc), C.c_int, _pd, _pd]
        f(C.byref(self._cdesc), q.shape[0], _p(q, _pd), _p(out, _pd))
        return out

    def eval_visual_tf(self, q):
        """visual_tf of every collision link, (nq, L, 4, 4) (gto/gto_models.py:83-101)."""
        q = _f64(q).reshape(-1, self.desc.ndof)
        F, L = self.desc.n_frames, self.desc.n_links
        fr = np.empty((F, 12))
        vis = np.empty((L, 12))
        out = np.zeros((q.shape[0], L, 4, 4))
        self.lib.orc_fk_affine.argtypes = [C.POINTER(CRobotDesc), _pd, _pd]
        self.lib.orc_visual_affine.argtypes = [C.POINTER(CRobotDesc), _pd, _pd]
        for i in range(q.shape[0]):
            self.lib.orc_fk_affine(C.byref(self._cdesc), _p(q[i], _pd), _p(fr, _pd))
            self.lib.orc_visual_affine(C.byref(self._cdesc), _p(fr, _pd), _p(vis, _pd))
            out[i, :, :3, :] = vis.reshape(L, 3, 4)
            out[i, :, 3, 3] = 1.0
        return out

    def eval_points(self, scene_id, q, base_pos, use_obs=False, want_field=True):
        q = _f64(q).reshape(-1, self.desc.ndof)
        nq, P = q.shape[0], self.desc.n_points
        base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (nq, 3)))
        xyz = np.empty((nq, P, 3))
        off = np.empty((nq, P), dtype=np.int32) if want_field else None
        val = np.empty((nq, P)) if want_field else None
        grad = np.empty((nq, P, 3)) if want_field else None
        sc = self._scenes[scene_id] if want_field else self._null_scene()
        f = self.lib.orc_eval_points
        f.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CScene), C.c_int32, _pd, _pd, C.c_int32, _pd, _pi, _pd, _pd]
        f(C.byref(self._cdesc), C.byref(sc), nq, _p(q, _pd), _p(base, _pd), int(use_obs), _p(xyz, _pd),
          _p(off, _pi), _p(val, _pd), _p(grad, _pd))
        return xyz, off, val, grad

    def _batch_args(self, B, scene_id, goals, n_goals, standoff, base_pos):
        goals = _f64(goals).reshape(B, -1, 16)
        n_goals = _i32(np.broadcast_to(np.asarray(n_goals), (B,)))
        scene_id = _i32(np.broadcast_to(np.asarray(scene_id), (B,)))
        so = None if standoff is None else _f64(np.broadcast_to(_f64(standoff).reshape(-1, 16), (B, 16)))
        base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (B, 3)))
        return goals, goals.shape[1], n_goals, scene_id, so, base

    def solve_batch(self, scene_id, qc, goals, n_goals, standoff, base_pos, Q0, n_threads=0, trace=False):
        d, T = self.desc, self.T
        qc = _f64(qc).reshape(-1, d.ndof)
        B = qc.shape[0]
        goals, n_max, n_goals, scene_id, so, base = self._batch_args(B, scene_id, goals, n_goals, standoff, base_pos)
        Q0 = _f64(Q0).reshape(B, d.ndof, T)
        Q, dQ = np.empty((B, d.ndof, T)), np.empty((B, d.ndof, T - 1))
        cost = np.empty(B)
        iters, status = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        ftr = np.full((B, self.opts.max_iter + 1), np.nan) if trace else None
        f = self.lib.orc_solve_batch
        f.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CSolverOpts), C.POINTER(CScene), C.c_int32, C.c_int32,
                      _pi, _pd, _pd, _pi, _pd, _pd, _pd, _pd, _pd, _pd, _pi, _pi, _pd, C.c_int32]
        f.restype = C.c_int
        rc = f(C.byref(self._cdesc), C.byref(self.opts), self._scene_array(), B, n_max, _p(scene_id, _pi),
               _p(qc, _pd), _p(goals, _pd), _p(n_goals, _pi), _p(so, _pd), _p(base, _pd), _p(Q0, _pd),
               _p(Q, _pd), _p(dQ, _pd), _p(cost, _pd), _p(iters, _pi), _p(status, _pi), _p(ftr, _pd), n_threads)
        if rc != 0:
            raise RuntimeError(f"orc_solve_batch failed ({rc})")
        if trace:
            return Q, dQ, cost, iters, status, ftr
        return Q, dQ, cost, iters, status

    def solve_ik_batch(self, scene_id, q0, goals, base_pos=None, max_iter=50, n_threads=0):
        """IK for B goal poses (gto/ik_solver.py:78-110).  scene_id None: no collision term.
        Returns (q (B,ndof), cost (B,), iters (B,), status (B,))."""
        d = self.desc
        q0 = _f64(q0).reshape(-1, d.ndof)
        B = q0.shape[0]
        goals = _f64(goals).reshape(B, 16)
        sid = None if scene_id is None else _i32(np.broadcast_to(np.asarray(scene_id), (B,)))
        base = _f64(np.broadcast_to(_f64([0, 0, 0] if base_pos is None else base_pos).reshape(-1, 3), (B, 3)))
        q, cost = np.empty((B, d.ndof)), np.empty(B)
        iters, status = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        opts = CSolverOpts.from_buffer_copy(self.opts)
        opts.max_iter = max_iter
        f = self.lib.orc_solve_ik_batch
        f.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CSolverOpts), C.POINTER(CScene), C.c_int32, _pi, _pd, _pd, _pd,
                      _pd, _pd, _pi, _pi, C.c_int32]
        f.restype = C.c_int
        rc = f(C.byref(self._cdesc), C.byref(opts), self._scene_array(), B, _p(sid, _pi), _p(q0, _pd), _p(goals, _pd),
               _p(base, _pd), _p(q, _pd), _p(cost, _pd), _p(iters, _pi), _p(status, _pi), n_threads)
        if rc != 0:
            raise RuntimeError(f"orc_solve_ik_batch failed ({rc})")
        return q, cost, iters, status

    def solve_base_batch(self, qc, goals, n_goals=None, effort_weight=0.01, max_iter=100, n_threads=0):
        """Base placement for B goal sets (gto/base_planner.py:35-134): qc (B,ndof), goals (B,n_max,4,4).
        Returns (y (B,3) = x, y, theta, q (B,n_max,ndof), cost (B,), iters (B,), status (B,))."""
        d = self.desc
        qc = _f64(qc).reshape(-1, d.ndof)
        B = qc.shape[0]
        goals = _f64(goals).reshape(B, -1, 16)
        n_max = goals.shape[1]
        ng = _i32(np.broadcast_to(np.asarray(n_max if n_goals is None else n_goals), (B,)))
        y, q, cost = np.empty((B, 3)), np.zeros((B, n_max, d.ndof)), np.empty(B)
        iters, status = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        f = self.lib.orc_solve_base_batch
        f.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CSolverOpts), C.c_int32, C.c_int32, _pi, _pd, _pd, C.c_double,
                      C.c_int32, _pd, _pd, _pd, _pi, _pi, C.c_int32]
        f.restype = C.c_int
        rc = f(C.byref(self._cdesc), C.byref(self.opts), B, n_max, _p(ng, _pi), _p(qc, _pd), _p(goals, _pd),
               float(effort_weight), int(max_iter), _p(y, _pd), _p(q, _pd), _p(cost, _pd), _p(iters, _pi),
               _p(status, _pi), n_threads)
        if rc != 0:
            raise RuntimeError(f"orc_solve_base_batch failed ({rc})")
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
        f = self.lib.orc_eval_base_objective
        f.argtypes = [C.POINTER(CRobotDesc), C.c_int32, C.c_int32, _pi, _pd, _pd, _pd, C.c_double, _pd]
        f.restype = C.c_int
        rc = f(C.byref(self._cdesc), B, n_max, _p(ng, _pi), _p(y, _pd), _p(q, _pd), _p(goals, _pd), float(effort_weight),
               _p(cost, _pd))
        if rc != 0:
            raise RuntimeError(f"orc_eval_base_objective failed ({rc})")
        return cost

    def eval_objective(self, scene_id, goals, n_goals, standoff, base_pos, Q):
        d, T = self.desc, self.T
        Q = _f64(Q).reshape(-1, d.ndof, T)
        B = Q.shape[0]
        goals, n_max, n_goals, scene_id, so, base = self._batch_args(B, scene_id, goals, n_goals, standoff, base_pos)
        fg, fo, fv = np.empty(B), np.empty(B), np.empty(B)
        am = np.empty(B, dtype=np.int32)
        f = self.lib.orc_eval_objective
        f.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CSolverOpts), C.POINTER(CScene), C.c_int32, C.c_int32,
                      _pi, _pd, _pi, _pd, _pd, _pd, _pd, _pd, _pd, _pi]
        f(C.byref(self._cdesc), C.byref(self.opts), self._scene_array(), B, n_max, _p(scene_id, _pi),
          _p(goals, _pd), _p(n_goals, _pi), _p(so, _pd), _p(base, _pd), _p(Q, _pd), _p(fg, _pd), _p(fo, _pd),
          _p(fv, _pd), _p(am, _pi))
        return fg, fo, fv, am

    def eval_normal_eq(self, scene_id, goals, n_goals, standoff, base_pos, Q):
        """Obstacle blocks (JtJ [B,T,n,n], Jtr [B,T,n], sumsq [B,T]) and goal blocks ([B,2,n,n], [B,2,n])."""
        d, T, n = self.desc, self.T, self.desc.n_opt
        Q = _f64(Q).reshape(-1, d.ndof, T)
        B = Q.shape[0]
        if goals is None:
            gl, n_max, ng, so = None, 1, None, None
            scene_id = _i32(np.broadcast_to(np.asarray(scene_id), (B,)))
            base = _f64(np.broadcast_to(_f64(base_pos).reshape(-1, 3), (B, 3)))
        else:
            gl, n_max, ng, scene_id, so, base = self._batch_args(B, scene_id, goals, n_goals, standoff, base_pos)
        JtJ, Jtr, ss = np.empty((B, T, n, n)), np.empty((B, T, n)), np.empty((B, T))
        Hg, gg = np.zeros((B, 2, n, n)), np.zeros((B, 2, n))
        f = self.lib.orc_eval_normal_eq
        f.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CSolverOpts), C.POINTER(CScene), C.c_int32, C.c_int32,
                      _pi, _pd, _pi, _pd, _pd, _pd, _pd, _pd, _pd, _pd, _pd]
        # the C side stores goal blocks with stride GTO_MAX_OPT internally but writes n x n here
        f(C.byref(self._cdesc), C.byref(self.opts), self._scene_array(), B, n_max, _p(scene_id, _pi),
          _p(gl, _pd), _p(ng, _pi), _p(so, _pd), _p(base, _pd), _p(Q, _pd), _p(JtJ, _pd), _p(Jtr, _pd),
          _p(ss, _pd), _p(Hg, _pd), _p(gg, _pd))
        return JtJ, Jtr, ss, Hg, gg

    def eval_obstacle_normal_eq(self, scene_id, base_pos, Q):
        JtJ, Jtr, ss, _, _ = self.eval_normal_eq(scene_id, None, None, None, base_pos, Q)
        return JtJ, Jtr, ss

    def plan_cost(self, scene_id, plans, base_pos):
        d, T = self.desc, self.T
        plans = _f64(plans).reshape(-1, d.ndof, T)
        n = plans.shape[0]
        cost, dist = np.empty(n), np.empty(n)
        f = self.lib.orc_plan_cost
        f.argtypes = [C.POINTER(CRobotDesc), C.POINTER(CScene), C.c_int32, C.c_int32, _pd, _pd, _pd, _pd]
        f(C.byref(self._cdesc), C.byref(self._scenes[scene_id]), n, T, _p(plans, _pd),
          _p(_f64(np.asarray(base_pos).reshape(3)), _pd), _p(cost, _pd), _p(dist, _pd))
        return cost, dist

    @staticmethod
    def num_threads():
        return lib().orc_num_threads()

    @staticmethod
    def usable_cores():
        """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container on a
        256-thread host may be limited to 16 CPUs' worth of time: more OpenMP threads than that only add contention)."""
        import math
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as fh:  # cgroup v2: "<quota|max> <period>"
                q, p = fh.read().split()
                if q != "max":
                    quota = float(q) / float(p)
        except (OSError, ValueError):
            try:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                    q, p = float(fq.read()), float(fp.read())
                    if q > 0:
                        quota = q / p
            except (OSError, ValueError):
                pass
        if quota is not None:
            n = min(n, max(1, int(math.floor(quota))))
        return max(1, n)
