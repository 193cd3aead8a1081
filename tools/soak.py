#!/usr/bin/env python3
"""Soak test: handles created and destroyed, scenes replaced and dropped, solves of random sizes in both modes, IK, base
placement and depth fields interleaved, for a fixed time; checks solver invariants on every result and that device memory
comes back (hipMemGetInfo through torch) when everything is closed.
usage: python tools/soak.py [seconds=60]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from grasptrajopt_amd import _capi, synthetic as syn  # noqa: E402
import grasptrajopt_amd as g  # noqa: E402
from helpers import Problem  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(0)
    torch.cuda.init()
    free0 = torch.cuda.mem_get_info()[0]
    t0, n_solves, n_inst = time.time(), 0, 0
    handles = []
    it = 0
    while time.time() - t0 < budget:
        it += 1
        robot = str(rng.choice(["panda", "fetch", "panda_5k"]))
        T = int(rng.choice([12, 30, 50]))
        B = int(rng.choice([1, 3, 17, 64, 200, 450]))
        prob = Problem(robot, B=B, scene_seed=int(rng.integers(0, 1000)), T=T, n_goals=int(rng.integers(1, 4)))
        opts = _capi.default_opts()
        opts.T, opts.standoff_offset, opts.max_iter = T, -max(2, T // 5), int(rng.choice([5, 30, 100]))
        h = _capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
        h.set_mode(0)
        # lanes inside the call (gto_set_lanes): any setting, same results as far as the invariants below can tell
        h.set_lanes(int(rng.integers(1, 9)), int(rng.choice([1, 8, 64, 256])), int(rng.choice([0, 3, 48, 1000])))
        prob.finish(h.eval_fk)
        for rep in range(int(rng.integers(1, 4))):  # replace the scene a few times (spare buffers), sometimes values-only ids
            h.set_scene(0, prob.scene.c_all, prob.scene.c_obs, prob.scene.shape, prob.scene.origin, prob.scene.res)
        h.set_scene(2, prob.scene.c_all, None, prob.scene.shape, prob.scene.origin, prob.scene.res, values_only=True)
        for rep in range(int(rng.integers(1, 4))):
            Q, dQ, f, iters, st = h.solve_batch(*prob.solve_args())
            n_solves += 1
            n_inst += B
            oi = prob.desc.opt_index
            assert np.isfinite(Q).all() and np.isfinite(f).all()
            assert (Q[:, oi] >= prob.desc.lower[oi][None, :, None] - 1e-9).all() and (Q[:, oi] <= prob.desc.upper[oi][None, :, None] + 1e-9).all()
            np.testing.assert_allclose(Q[:, :, 0], prob.qc, atol=0)
            np.testing.assert_allclose(Q[:, :, 1], prob.qc, atol=0)
            fg, fo, fv, _ = h.eval_objective(0, prob.goals, prob.n_goals, prob.S, prob.base, Q)
            np.testing.assert_allclose(fg + fo + fv, f, rtol=1e-9)
            assert ((iters >= 0) & (iters <= opts.max_iter)).all() and np.isin(st, [0, 1]).all()
        if rng.random() < 0.5:
            h.solve_ik_batch(0, prob.qc, prob.goals[:, 0], prob.base, max_iter=20)
        if rng.random() < 0.3:
            h.plan_cost(2, prob.Q0, prob.base[0])
        if rng.random() < 0.3:
            h.drop_scene(2)
        if rng.random() < 0.2:
            depth = (0.7 + 0.3 * rng.random((60, 80))).astype(np.float32)
            cam = np.eye(4); cam[:3, 3] = [0, 0, 1.0]
            dpc = g.DepthPointCloud(depth, np.array([[70.0, 0, 40], [0, 70.0, 30], [0, 0, 1]]), cam)
            dpc.get_sdf_cost(rng.uniform(-1, 1, size=(5000, 3)))
        handles.append(h)
        if len(handles) > 3 or rng.random() < 0.5:  # close in random order
            k = int(rng.integers(0, len(handles)))
            handles.pop(k).close()
    for h in handles:
        h.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print(f"{it} rounds, {n_solves} solve calls, {n_inst} instances in {time.time()-t0:.0f} s; device memory not returned: "
          f"{(free0 - free1) / 2**20:.1f} MiB (one-time: code objects, kernel scratch, the depth-field pool; tools/leak_check.py shows "
          "that it does not grow with the number of cycles)")


if __name__ == "__main__":
    main()
