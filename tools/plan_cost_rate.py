#!/usr/bin/env python3
"""Seed scoring rate (row f-3, gto_plan_cost = GTORobotModel.compute_plan_cost for n plans at once).
usage: tools/plan_cost_rate.py [n_plans]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grasptrajopt_amd import _capi, synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
desc = load_builtin("panda_5k")
opts = _capi.default_opts()
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
sc = syn.make_scene(0, n=128, res=2.24 / 128, origin=(-0.4, -1.12, -0.4), table_z=-0.03)
h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
rng = np.random.default_rng(0)
lo, hi = desc.lower[desc.opt_index], desc.upper[desc.opt_index]
Q = np.zeros((n, desc.ndof, opts.T))
Q[:, desc.opt_index, :] = rng.uniform(lo[None, :, None], hi[None, :, None], (n, len(lo), opts.T))
h.plan_cost(0, Q[:8], [0, 0, 0])
t = time.perf_counter()
for _ in range(3):
    c, d = h.plan_cost(0, Q, [0, 0, 0])
dt = (time.perf_counter() - t) / 3
print(f"{n} plans x {opts.T} waypoints x {desc.n_points} points: {dt*1e3:.2f} ms per call incl. H2D/D2H ({n/dt/1e3:.1f} k plans/s, {n*opts.T*desc.n_points/dt/1e9:.1f} G point lookups/s); checksum {c.sum():.6f}")
