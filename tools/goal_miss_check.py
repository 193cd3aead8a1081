#!/usr/bin/env python3
"""Are the goal-pose misses of the bench's quality block (goal_ok_frac ~ 0.78 against the IK thresholds 1 cm / 5 deg)
a property of the reference's objective or of this solver?  On the smooth part of the problem (goal-set matching +
standoff + velocity smoothness under the joint limits; empty cost field, i.e. what IPOPT sees gradient-wise) the CPU
oracle's projected LM and SciPy's L-BFGS-B (third party) are run from the same seeds with the reference's weights
(T = 50, w_vel = 0.01, Tmax = 10) and the end-effector pose error at BOTH minimisers is reported.
usage: python tools/goal_miss_check.py [n_instances]      (CPU only, ~10 s per instance)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pose_err(o, desc, cfg, Q, RT):
    Tf = o.eval_fk(Q[:, :, -1])[:, desc.frame_index(cfg["link_ee"])]
    ep = np.linalg.norm(Tf[:, :3, 3] - RT[:, :3, 3], axis=1)
    ca = (np.einsum("bij,bij->b", Tf[:, :3, :3], RT[:, :3, :3]) - 1.0) / 2.0
    return ep, np.degrees(np.arccos(np.clip(ca, -1, 1)))


def main():
    from helpers import Problem
    from independent import SmoothProblem
    from oracle import oracle
    oracle.build()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    prob = Problem("panda", B=B, scene_seed=0, T=50)
    opts = oracle.reference_opts(grad_mode=1, max_iter=300, tol_rel_f=1e-13)
    o = oracle.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(o.eval_fk)
    zero = np.zeros_like(prob.scene.c_all)
    o.set_scene(0, zero, zero, prob.scene.shape, prob.scene.origin, prob.scene.res)
    Q, _, f, it, st = o.solve_batch(*prob.solve_args())
    RT = prob.goals[:, 0].reshape(B, 4, 4)
    ep, er = pose_err(o, prob.desc, prob.cfg, Q, RT)
    Qs = Q.copy()
    fs = np.zeros(B)
    for b in range(B):
        sp = SmoothProblem(o, prob, b, opts, Q[b])
        res = sp.lbfgsb(sp.pack(prob.Q0[b]))
        Qs[b] = sp.unpack(res.x)
        fs[b] = res.fun
    eps, ers = pose_err(o, prob.desc, prob.cfg, Qs, RT)
    print("inst |  LM: f         pos err m  rot err deg | L-BFGS-B: f     pos err m  rot err deg | max |dQ| rad")
    for b in range(B):
        print(f"{b:4d} | {f[b]:12.9f} {ep[b]:9.5f} {er[b]:9.3f}   | {fs[b]:12.9f} {eps[b]:9.5f} {ers[b]:9.3f}   | {np.abs(Q[b] - Qs[b]).max():.2e}")
    ok = lambda p, r: ((p < 0.01) & (r < 5.0)).mean()
    print(f"goal_ok_frac (1 cm, 5 deg): LM {ok(ep, er):.3f}   L-BFGS-B {ok(eps, ers):.3f}   same basin (|dQ| < 1e-3): {(np.abs(Q - Qs).reshape(B, -1).max(1) < 1e-3).mean():.2f}")


if __name__ == "__main__":
    main()
