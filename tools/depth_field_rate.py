#!/usr/bin/env python3
"""depth_field_rate.py — time of one cost-field construction (gto_depth_sdf_cost) at the reference's sizes:
a 480x640 depth image and the voxel centres of the Panda workspace grid at 5 cm, then at 128^3."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grasptrajopt_amd as g  # noqa: E402

rng = np.random.default_rng(0)
H, W = 480, 640
K = np.array([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1.0]])
noise = len(sys.argv) > 1 and sys.argv[1] == "noise"
if noise:  # every pixel at a random depth: the worst case for the bounding boxes of the image tiles
    depth = (0.8 + 0.4 * rng.random((H, W))).astype(np.float32)
else:  # a floor seen at an angle with a few boxes on it, one millimetre of sensor noise
    v, u = np.mgrid[0:H, 0:W]
    depth = (1.0 + 0.0012 * (v - H / 2) + 0.0003 * (u - W / 2)).astype(np.float32)
    for (r0, r1, c0, c1, dz) in ((150, 260, 200, 330, 0.2), (280, 400, 380, 520, 0.1), (100, 180, 420, 480, 0.3)):
        depth[r0:r1, c0:c1] -= dz
    depth += (0.001 * rng.standard_normal((H, W))).astype(np.float32)
a = 0.5
cam = np.eye(4)
cam[:3, :3] = np.array([[0, -np.sin(a), np.cos(a)], [-1.0, 0, 0], [0, -np.cos(a), -np.sin(a)]])
cam[:3, 3] = [-0.3, 0.0, 0.9]
dpc = g.DepthPointCloud(depth, K, cam)
for n in (48, 128):
    ax = np.linspace(-0.4, 1.84, n)
    q = np.stack(np.meshgrid(ax, ax - 0.72, ax, indexing="ij"), -1).reshape(-1, 3)
    dpc.get_sdf_cost(q[:1000])
    t0 = time.perf_counter()
    c = dpc.get_sdf_cost(q)
    dt = time.perf_counter() - t0
    print(f"{H}x{W} depth ({(depth > 0).sum()} points), {n}^3 = {q.shape[0]} voxels: {1e3 * dt:8.1f} ms  "
          f"({q.shape[0] * H * W / dt / 1e9:.1f} G point-pairs/s), non-zero cost voxels {int((c > 0).sum())}")
