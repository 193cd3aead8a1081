"""Seeded synthetic problem instances for benchmarking and parity tests.

No SceneReplica data is available (reference .gitignore:1), so scenes are a table slab plus a few
random boxes/spheres voxelised on a grid that covers the arm's reach box; the signed distance is
mapped to the reference's cost exactly as DepthPointCloud.get_sdf_cost does
(mesh_to_sdf/depth_point_cloud.py:84-89, float32 arithmetic), and ``c_obs`` is the same field with
the target object removed (examples/pybullet_gto_planning.py:181-190).
This is input generation (host, numpy); it is not part of the solve path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple, Sequence

import numpy as np


@dataclass
class Scene:
    c_all: np.ndarray    # (N0*N1*N2,) float32, C order (x slowest)
    c_obs: np.ndarray
    shape: Tuple[int, int, int]
    origin: np.ndarray   # (3,)
    res: float
    objects: list


def sdf_cost_map(signed_dist: np.ndarray, epsilon: float = 0.02, w_inside: float = 1.0) -> np.ndarray:
    """mesh_to_sdf/depth_point_cloud.py:84-89 on float32 arrays (inside <=> negative distance)."""
    d = signed_dist.astype(np.float32)
    cost = np.zeros_like(d)
    inside = d < 0
    cost[inside] = np.float32(w_inside) * (-d[inside] + np.float32(epsilon) / np.float32(2))
    idx = (d > 0) & (d < np.float32(epsilon))
    cost[idx] = np.square(d[idx] - np.float32(epsilon)) / np.float32(2 * epsilon)
    return cost


def _box_sdf(p, c, h):
    q = np.abs(p - c) - h
    return np.linalg.norm(np.maximum(q, 0.0), axis=-1) + np.minimum(q.max(axis=-1), 0.0)


def _sphere_sdf(p, c, r):
    return np.linalg.norm(p - c, axis=-1) - r


def make_scene(seed: int, n: int = 128, res: float = 0.0175,
               origin=(-0.4, -1.12, -0.4), table_z: float = -0.03, shelf: bool = False) -> Scene:
    """Table slab below ``table_z`` + K in [3,8] random boxes/spheres standing on it.  ``shelf``: the objects stand in a
    shelf compartment instead (SceneReplica shelf scenes): the lower board is the support at ``table_z`` (2 cm thick, no
    slab below it), an upper board 0.40 m above it, a back wall and two side walls around x in [0.35, 0.95],
    y in [-0.6, 0.6]; the compartment is open towards the robot."""
    rng = np.random.default_rng(1000 + seed)
    origin = np.asarray(origin, dtype=np.float64)
    ax = [origin[a] + res * np.arange(n) for a in range(3)]
    k = int(rng.integers(3, 9))
    objects = []
    boards = []
    if shelf:
        x0, x1, y0, y1, hgt, th = 0.35, 0.95, -0.6, 0.6, 0.40, 0.01
        xc, xh, yc, yh = (x0 + x1) / 2, (x1 - x0) / 2, (y0 + y1) / 2, (y1 - y0) / 2
        boards = [(np.array([xc, yc, table_z - th]), np.array([xh, yh, th])),                    # lower board (support)
                  (np.array([xc, yc, table_z + hgt + th]), np.array([xh, yh, th])),             # upper board
                  (np.array([x1 + th, yc, table_z + hgt / 2]), np.array([th, yh, hgt / 2 + 2 * th])),  # back wall
                  (np.array([xc, y0 - th, table_z + hgt / 2]), np.array([xh, th, hgt / 2 + 2 * th])),  # side walls
                  (np.array([xc, y1 + th, table_z + hgt / 2]), np.array([xh, th, hgt / 2 + 2 * th]))]
    for i in range(k):
        cx, cy = rng.uniform(0.3, 0.8), rng.uniform(-0.5, 0.5)
        if rng.random() < 0.6:
            h = rng.uniform([0.03, 0.03, 0.04], [0.08, 0.08, 0.15])
            objects.append(("box", np.array([cx, cy, table_z + h[2]]), h))
        else:
            r = rng.uniform(0.03, 0.07)
            objects.append(("sphere", np.array([cx, cy, table_z + r]), r))
    target = int(rng.integers(0, k))

    def field(skip: Optional[int]) -> np.ndarray:
        """Signed distance, exact wherever it is below `far` (the cost map is zero from epsilon = 2 cm on, so the distance
        to an object is only evaluated in the voxels of its bounding box grown by `far`; elsewhere the object cannot be
        the nearest surface within the cost band).  Same arithmetic per voxel as evaluating every object everywhere."""
        far = 0.05
        if shelf:
            out = np.full((n, n, n), np.inf, dtype=np.float64)
        else:
            out = np.broadcast_to(ax[2] - table_z, (n, n, n)).copy()  # half-space z < table_z

        def stamp(kind, c, sz):
            half = np.asarray(sz, dtype=np.float64) if kind == "box" else np.full(3, float(sz))
            lo = [int(np.searchsorted(ax[a_], c[a_] - half[a_] - far, side="left")) for a_ in range(3)]
            hi = [int(np.searchsorted(ax[a_], c[a_] + half[a_] + far, side="right")) for a_ in range(3)]
            if any(l_ >= h_ for l_, h_ in zip(lo, hi)):
                return
            X, Y, Z = np.meshgrid(ax[0][lo[0]:hi[0]], ax[1][lo[1]:hi[1]], ax[2][lo[2]:hi[2]], indexing="ij")
            p = np.stack([X, Y, Z], axis=-1)
            d = _box_sdf(p, c, sz) if kind == "box" else _sphere_sdf(p, c, sz)
            sub = out[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
            np.minimum(sub, d, out=sub)

        for c_, h_ in boards:
            stamp("box", c_, h_)
        for j, (kind, c, sz) in enumerate(objects):
            if j != skip:
                stamp(kind, c, sz)
        return out.astype(np.float32)

    c_all = sdf_cost_map(field(None)).reshape(-1)
    c_obs = sdf_cost_map(field(target)).reshape(-1)
    return Scene(c_all=c_all, c_obs=c_obs, shape=(n, n, n), origin=origin, res=res,
                 objects=[(kd, c.tolist(), (s.tolist() if kd == "box" else float(s))) for kd, c, s in objects]
                 + [("target", target)] + ([("shelf", [[c_.tolist(), h_.tolist()] for c_, h_ in boards])] if shelf else []))


def standoff_pose(offset: float, axis: str) -> np.ndarray:
    """optas/spatialmath.py:160-183 standoff()."""
    S = np.eye(4)
    S["xyz".index(axis), 3] = offset
    return S


def interpolate_waypoints(waypoints: np.ndarray, n: int, m: int) -> np.ndarray:
    """gto/utils.py:63-82 for the planner's two-waypoint call: clamped cubic through the two
    configurations, sampled at linspace(0,1,n+2)[1:-1].  Returns (n, m)."""
    w = np.asarray(waypoints, dtype=np.float64)
    if w.shape[0] != 2:
        raise NotImplementedError("only the two-waypoint call made by GTOPlanner is supported")
    s = np.arange(1, n + 1, dtype=np.float64) / (n + 1)
    h = s * s * (3.0 - 2.0 * s)
    return w[0][None, :] + (w[1] - w[0])[None, :] * h[:, None]


def make_goals(desc, fk: Callable[[np.ndarray], np.ndarray], link_ee: str, n_goals: int, seed: int,
               xlim=(0.25, 0.75), ylim=(-0.5, 0.5), zlim=(0.08, 0.7),
               collision_cost: Optional[Callable[[np.ndarray], np.ndarray]] = None):
    """Sample in-limit configurations q* whose end-effector lies in a box above the table;
    goal pose RT = FK(q*).  ``fk(q) -> (nq, n_frames, 4, 4)``.  ``collision_cost(q) -> (nq,)`` (optional)
    rejects configurations whose moving links touch the obstacle band (the reference's pipeline only
    passes collision-free IK solutions on, examples/pybullet_gto_planning.py:207-263).
    Returns (RT (n,4,4), q* (n,ndof))."""
    rng = np.random.default_rng(5000 + seed)
    lo = np.maximum(desc.lower, -3.0)
    hi = np.minimum(desc.upper, 3.0)
    fe = desc.frame_index(link_ee)
    RTs, qs = [], []
    while len(RTs) < n_goals:
        q = rng.uniform(lo, hi, size=(4 * n_goals, desc.ndof))
        q[:, desc.param_index] = 0.0
        T = fk(q)[:, fe]
        p = T[:, :3, 3]
        ok = ((p[:, 0] > xlim[0]) & (p[:, 0] < xlim[1]) & (p[:, 1] > ylim[0]) & (p[:, 1] < ylim[1])
              & (p[:, 2] > zlim[0]) & (p[:, 2] < zlim[1]))
        if collision_cost is not None and ok.any():
            idx = np.nonzero(ok)[0]
            ok[idx[collision_cost(q[idx]) > 0.0]] = False
        for i in np.nonzero(ok)[0]:
            if len(RTs) < n_goals:
                RTs.append(T[i])
                qs.append(q[i])
    return np.stack(RTs), np.stack(qs)


# Joint-space distance between the first and the last configuration of the reference's 186 stored Panda table-top plans
# (examples/results_iros2024/GTO_scenereplica_panda_tabletop_*.json; tests/golden/plan_statistics.npz `chord`, made by
# tests/golden/make_plan_statistics.py): percentiles 0, 5, 10, 25, 50, 75, 90, 95, 100 in rad; Fetch: 184 plans.
STORED_CHORD_PCT = (0.0, 5.0, 10.0, 25.0, 50.0, 75.0, 90.0, 95.0, 100.0)
STORED_CHORD_RAD = {"panda": (1.381, 1.637, 1.717, 1.878, 2.166, 2.570, 3.359, 3.482, 4.203),
                    "fetch": (1.812, 2.257, 2.450, 2.666, 2.925, 3.314, 3.623, 3.915, 4.505)}


def make_goal_sets_reference_shaped(desc, fk: Callable[[np.ndarray], np.ndarray], link_ee: str, qc, n_sets: int, set_size: int, seed: int,
                                    chord_rad: Sequence[float], xlim=(0.25, 0.75), ylim=(-0.5, 0.5), zlim=(0.08, 0.7), spread: float = 0.25,
                                    collision_cost: Optional[Callable[[np.ndarray], np.ndarray]] = None):
    """Goal sets shaped like what the reference's driver hands to plan_goalset (examples/pybullet_gto_planning.py:242-293):
    `set_size` collision-free IK solutions of grasps of ONE object -- here configurations within `spread` rad (per joint,
    normal) of a set's first configuration, end effector in the same box above the table -- whose first configuration lies
    at a joint-space distance from qc drawn from the stored plans' distribution (`chord_rad`: values at STORED_CHORD_PCT).
    The synthetic goals of make_goals are uniform over the joint limits: 2.9-5.0 rad away, the stored plans 1.6-3.5.
    Returns (RT (n_sets, set_size, 4, 4), q (n_sets, set_size, ndof))."""
    rng = np.random.default_rng(9000 + seed)
    qc = np.asarray(qc, dtype=np.float64)
    oi = np.asarray(desc.opt_index)
    lo, hi = np.asarray(desc.lower)[oi], np.asarray(desc.upper)[oi]
    fe = desc.frame_index(link_ee)

    def admissible(q):
        T = np.asarray(fk(q))[:, fe]
        p = T[:, :3, 3]
        ok = ((p[:, 0] > xlim[0]) & (p[:, 0] < xlim[1]) & (p[:, 1] > ylim[0]) & (p[:, 1] < ylim[1]) & (p[:, 2] > zlim[0]) & (p[:, 2] < zlim[1]))
        if collision_cost is not None and ok.any():
            idx = np.nonzero(ok)[0]
            ok[idx[collision_cost(q[idx]) > 0.0]] = False
        return ok, T

    RT, Qs = np.zeros((n_sets, set_size, 4, 4)), np.tile(qc, (n_sets, set_size, 1))
    for b in range(n_sets):
        for _ in range(200):
            c = float(np.interp(100.0 * rng.random(), STORED_CHORD_PCT, chord_rad))
            u = rng.standard_normal((64, len(oi)))
            q = np.tile(qc, (64, 1))
            q[:, oi] = np.clip(qc[oi] + c * u / np.linalg.norm(u, axis=1, keepdims=True), lo, hi)
            ok, T = admissible(q)
            ok &= np.abs(np.linalg.norm(q[:, oi] - qc[oi], axis=1) - c) < 0.05  # (a clipped direction is shorter)
            if not ok.any():
                continue
            i0 = int(np.nonzero(ok)[0][0])
            members, poses = [q[i0]], [T[i0]]
            for _ in range(50):
                qq = np.tile(q[i0], (32, 1))
                qq[:, oi] = np.clip(q[i0][oi] + spread * rng.standard_normal((32, len(oi))), lo, hi)
                ok2, T2 = admissible(qq)
                for j in np.nonzero(ok2)[0]:
                    if len(members) < set_size:
                        members.append(qq[j])
                        poses.append(T2[j])
                if len(members) == set_size:
                    break
            if len(members) == set_size:
                RT[b], Qs[b] = np.stack(poses), np.stack(members)
                break
        else:
            raise RuntimeError("make_goal_sets_reference_shaped: no admissible goal set found")
    return RT, Qs


def make_seed(qc: np.ndarray, q_goal: np.ndarray, T: int, param_index) -> np.ndarray:
    """Seed trajectory as GTOPlanner.plan builds it (gto/gto_planner.py:155-158): interpolate
    qc -> q_goal, parameter joints held at qc.  Returns (ndof, T)."""
    data = interpolate_waypoints(np.stack([qc, q_goal]), T, qc.shape[0])
    data[:, param_index] = np.asarray(qc)[param_index]
    return data.T.copy()


def base_pose_matrix(y) -> np.ndarray:
    """rt2tr(rotz(theta), [x, y, 0]) of a planar base pose y = (x, y, theta) (gto/base_planner.py:49-51)."""
    c, s = np.cos(y[2]), np.sin(y[2])
    M = np.eye(4)
    M[:2, :2] = [[c, -s], [s, c]]
    M[0, 3], M[1, 3] = y[0], y[1]
    return M


def make_base_goal_sets(desc, fk: Callable[[np.ndarray], np.ndarray], link_ee: str, qc, B: int, n: int, seed: int,
                        spread: float = 0.5, shift: float = 0.4, turn: float = 0.6):
    """Goal sets for the base-placement problem: every set is reachable from ONE displaced base pose y*
    (|x|,|y| <= shift, |theta| <= turn): arm configurations within `spread` of qc, goal_i = B(y*)^-1 FK_ee(q_i),
    i.e. poses of link_ee seen from the current base.  Returns (goals (B,n,4,4), y* (B,3))."""
    rng = np.random.default_rng(7000 + seed)
    qc = np.asarray(qc, dtype=np.float64)
    oi = np.asarray(desc.opt_index)
    lo, hi = np.asarray(desc.lower)[oi], np.asarray(desc.upper)[oi]
    fe = desc.frame_index(link_ee)
    goals, ystar = np.zeros((B, n, 4, 4)), np.zeros((B, 3))
    for b in range(B):
        ystar[b] = [rng.uniform(-shift, shift), rng.uniform(-shift, shift), rng.uniform(-turn, turn)]
        Binv = np.linalg.inv(base_pose_matrix(ystar[b]))
        q = np.tile(qc, (n, 1))
        q[:, oi] = np.clip(qc[oi] + rng.uniform(-spread, spread, size=(n, len(oi))), lo, hi)
        goals[b] = Binv @ np.asarray(fk(q))[:, fe]
    return goals, ystar
