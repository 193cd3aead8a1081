#!/usr/bin/env python3
"""tg_identity.py — are solved trajectories bit-identical across waypoint-group sizes (GTO_OBS_TG)?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from helpers import Problem
    from grasptrajopt_amd import _capi
    prob = Problem("panda_5k", B=48, scene_seed=3, n=96, res=2.24 / 96)
    out = {}
    for tg in (1, 2, 3, 4, 6):
        os.environ["GTO_OBS_TG"] = str(tg)
        h = _capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], _capi.default_opts(), device=0)
        if "goals" not in prob.__dict__:
            prob.finish(h.eval_fk)
        h.set_scene(*prob.scene_args())
        out[tg] = h.solve_batch(*prob.solve_args())
        h.close()
    ref = out[2]
    for tg, r in out.items():
        print(tg, "Q identical" if np.array_equal(r[0], ref[0]) else f"max|dQ| {np.abs(r[0] - ref[0]).max():.3e}",
              "iters identical" if np.array_equal(r[3], ref[3]) else f"iters differ in {(r[3] != ref[3]).sum()}", "iters mean", r[3].mean())


if __name__ == "__main__":
    main()
