#!/usr/bin/env python3
"""ik_rate.py — throughput of the batched IK (gto_solve_ik_batch) on the bench robot/scene.
Usage: python tools/ik_rate.py [B]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grasptrajopt_amd import _capi, synthetic as syn  # noqa: E402
from grasptrajopt_amd.robot_desc import load_builtin  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
desc = load_builtin("panda_5k")
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], _capi.default_opts(), device=0, n_gripper_points=100)
sc = syn.make_scene(0, n=128, res=2.24 / 128)
h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
RT, qg = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=0)
q0 = np.tile(np.array(cfg["default_pose"]), (B, 1))
for collide in (None, 0):
    h.solve_ik_batch(collide, q0, RT.reshape(B, 16), np.zeros((B, 3)))
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        q, f, it, st = h.solve_ik_batch(collide, q0, RT.reshape(B, 16), np.zeros((B, 3)))
    dt = (time.perf_counter() - t0) / reps
    fe = desc.frame_index(cfg["link_ee"])
    Tf = h.eval_fk(q)[:, fe]
    ep = np.linalg.norm(Tf[:, :3, 3] - RT[:, :3, 3], axis=1)
    print(f"collision term {'on ' if collide is not None else 'off'}: B={B}  {1e3 * dt:7.2f} ms per call  {B / dt:9.0f} IK/s  "
          f"iters mean {it.mean():.1f} max {it.max()}  reached (<1 cm) {np.mean(ep < 0.01):.2f}")
