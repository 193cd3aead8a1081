#!/usr/bin/env python3
"""Device memory across create / set_scene / solve / destroy cycles (hipMemGetInfo through torch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from grasptrajopt_amd import _capi
from helpers import Problem
torch.cuda.init()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
prob = Problem("panda", B=64, scene_seed=1)
opts = _capi.default_opts(); opts.max_iter = 10
free = lambda: torch.cuda.mem_get_info()[0] / 2**20
f0 = free()
for i in range(60):
    h = _capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    if i == 0: prob.finish(h.eval_fk)
    if what in ("all", "scene", "solve", "mode1", "ik"):
        h.set_scene(0, prob.scene.c_all, prob.scene.c_obs, prob.scene.shape, prob.scene.origin, prob.scene.res)
        h.set_scene(0, prob.scene.c_all, prob.scene.c_obs, prob.scene.shape, prob.scene.origin, prob.scene.res)
    if what in ("all", "solve"):
        h.solve_batch(*prob.solve_args())
    if what in ("all", "ik"):
        h.solve_ik_batch(0, prob.qc, prob.goals[:, 0], prob.base, max_iter=5)
    h.close()
    if i % 10 == 9:
        torch.cuda.synchronize()
        print(f"{what}: after {i+1} cycles: {f0 - free():.1f} MiB not returned")
