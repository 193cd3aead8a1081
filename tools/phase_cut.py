#!/usr/bin/env python3
"""phase_cut.py — where does k_obstacle_gram spend its time and its instructions when the GPU is full?

Solves a batch (a normal handle), then runs ONE all-waypoints-active evaluation (gto_eval_obstacle_normal_eq) of the SOLVED
trajectories REPS times on a second handle whose obstacle kernel is cut short after a phase (GTO_DEBUG_CUT: 7 table staging,
8 sin/cos, 1 kinematics, 2 broad phase, 3 gather loop, 0 whole kernel; results of a cut kernel are garbage) and prints the
launch duration.  Under `rocprofv3 --kernel-trace --pmc ...` the per-dispatch counters of those launches give the
instructions per phase (tools/phase_cut_pmc.sh).
usage: python tools/phase_cut.py [--robot panda_5k|fetch] [--shelf] [--B 384] [--cut k] [--reps 6]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grasptrajopt_amd import _capi, synthetic as syn  # noqa: E402
from grasptrajopt_amd.robot_desc import load_builtin  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--robot", default="panda_5k")
ap.add_argument("--shelf", action="store_true")
ap.add_argument("--B", type=int, default=384)
ap.add_argument("--cut", type=int, default=int(os.environ.get("GTO_DEBUG_CUT", "0")))
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--max-iter", type=int, default=12, help="iterations of the solve whose end point is evaluated (mid-solve trajectories)")
a = ap.parse_args()
fetch = a.robot.startswith("fetch")
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", f"{a.robot.split('_')[0]}_cfg.json")))
desc = load_builtin(a.robot)
opts = _capi.default_opts()
opts.max_iter = a.max_iter
B, T = a.B, opts.T
os.environ.pop("GTO_DEBUG_CUT", None)
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
origin = (-0.3, -1.12, 0.0) if fetch else (-0.4, -1.12, -0.4)
table_z = 0.75 if a.shelf else (0.45 if fetch else -0.03)
sc = syn.make_scene(0, n=128, res=2.24 / 128, origin=origin, table_z=table_z, shelf=a.shelf)
h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
moving = desc.link_is_moving()[desc.point_link]


def cc(q):
    _, _, val, _ = h.eval_points(0, q, [0.0, 0.0, 0.0], use_obs=True)
    return (val * moving[None, :]).sum(axis=1)


zlim = (table_z + 0.07, table_z + 0.33) if a.shelf else ((0.55, 1.2) if fetch else (0.08, 0.7))
RT, qg = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=0, collision_cost=cc, zlim=zlim)
ndof = desc.ndof
qc = np.concatenate([np.zeros(ndof - len(cfg["default_pose"])), np.array(cfg["default_pose"], dtype=np.float64)])
Q0 = np.stack([syn.make_seed(qc, qg[b], T, desc.param_index) for b in range(B)])
if a.shelf:
    hold = np.repeat(np.tile(qc, (B, 1))[:, :, None], T, axis=2)
    hold[:, :, T + opts.standoff_offset:] = Q0[:, :, -1:]
    Q0 = hold
S = syn.standoff_pose(-0.1, cfg["axis_standoff"])
Q, _, _, it, _ = h.solve_batch(0, np.tile(qc, (B, 1)), RT.reshape(B, 1, 16), 1, S, [0, 0, 0], Q0)
if a.cut:
    os.environ["GTO_DEBUG_CUT"] = str(a.cut)
h2 = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
h2.share_scene(0, h)
h2.set_profiling(True)
ts = []
for _ in range(a.reps):
    h2.eval_obstacle_normal_eq(0, np.zeros((B, 3)), Q)
    ts.append(h2.last_kernel_time()[0])
pts, _ = h2.last_kernel_work()
print(f"cut {a.cut}  {a.robot}{' shelf' if a.shelf else ''} B {B} (iters mean {it.mean():.1f}): k_obstacle_gram {1e3 * min(ts[1:]):.1f} us (min of {a.reps - 1}), "
      f"per (instance, waypoint) {1e6 * min(ts[1:]) / (B * (T - 2)):.2f} ns")
h2.close()
h.close()
