#!/usr/bin/env python3
"""Time of gto_set_scene / gto_set_scene_values for a 128^3 scene, from the same arrays every time and from fresh ones."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grasptrajopt_amd import _capi, synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
desc = load_builtin("panda_5k")
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], _capi.default_opts(), device=0, n_gripper_points=100)
sc = syn.make_scene(0, n=128, res=0.0175)
for fresh in (False, True):
    for vo in (False, True):
        ts = []
        for r in range(12):
            ca, co = (sc.c_all + np.float32(1e-6 * r), sc.c_obs + np.float32(1e-6 * r)) if fresh else (sc.c_all, sc.c_obs)
            t = time.perf_counter()
            h.set_scene(0, ca, co, sc.shape, sc.origin, sc.res, values_only=vo)
            ts.append(time.perf_counter() - t)
        print(f"fresh arrays {fresh!s:5} values_only {vo!s:5} set_scene ms: median {np.median(ts[2:])*1e3:.2f} min {min(ts[2:])*1e3:.2f} max {max(ts[2:])*1e3:.2f}")
