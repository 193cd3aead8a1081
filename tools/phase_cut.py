#!/usr/bin/env python3
"""phase_cut.py — where does k_obstacle_gram spend its time when the GPU is full?

Runs ONE all-waypoints-active evaluation (gto_eval_obstacle_normal_eq) of a large batch with the kernel
cut short after its prologue / broad phase / gather loop (GTO_DEBUG_CUT, results are garbage) and
prints the launch duration.  Usage: GTO_DEBUG_CUT=k python tools/phase_cut.py [B]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grasptrajopt_amd import _capi, synthetic as syn  # noqa: E402
from grasptrajopt_amd.robot_desc import load_builtin  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
desc = load_builtin("panda_5k")
opts = _capi.default_opts()
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
sc = syn.make_scene(0, n=128, res=2.24 / 128)
h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
RT, qg = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=0)
qc = np.array(cfg["default_pose"])
Q0 = np.stack([syn.make_seed(qc, qg[b], opts.T, desc.param_index) for b in range(B)])
h.set_profiling(True)
ts = []
for _ in range(6):
    h.eval_obstacle_normal_eq(0, np.zeros((B, 3)), Q0)
    ts.append(h.last_kernel_time()[0])
print(f"cut {os.environ.get('GTO_DEBUG_CUT', '0')}  B {B}: k_obstacle_gram {1e3 * min(ts[1:]):.1f} us (min of 5), per (instance,waypoint) {1e6 * min(ts[1:]) / (B * 48):.2f} ns")
