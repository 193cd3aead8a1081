#!/usr/bin/env python3
"""parity_sweep.py — GPU vs CPU oracle over many seeded problems (scenes, robots, goal-set sizes, horizons,
gradient modes): iteration counts and status must be identical, trajectories within 1e-6 rad.
Usage: python tools/parity_sweep.py [--seeds 10] [--batch 32]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from helpers import Problem
    from grasptrajopt_amd import _capi
    from oracle import oracle
    nthr = oracle.Oracle.usable_cores()
    worst, bad, total = 0.0, 0, 0
    t0 = time.time()
    for seed in range(args.seeds):
        for robot, n_goals, T, off, grad in (("panda", 1, 50, -10, 0), ("fetch", 3, 50, -10, 0), ("panda_5k", 2, 30, -6, 0),
                                             ("panda", 4, 50, -10, 1), ("fetch", 1, 64, -12, 0)):
            prob = Problem(robot, B=args.batch, scene_seed=seed, n_goals=n_goals, T=T)
            opts = oracle.reference_opts(T=T, standoff_offset=off, grad_mode=grad)
            h = _capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
            o = oracle.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
            prob.finish(h.eval_fk)
            h.set_scene(*prob.scene_args())
            o.set_scene(*prob.scene_args())
            Qg, _, fg, itg, stg = h.solve_batch(*prob.solve_args())
            Qo, _, fo, ito, sto = o.solve_batch(*prob.solve_args(), n_threads=nthr)
            dq = np.abs(Qg - Qo).reshape(args.batch, -1).max(axis=1)
            same = (itg == ito) & (stg == sto)
            worst = max(worst, float(dq[same].max()) if same.any() else 0.0)
            bad += int((~same).sum()) + int((dq[same] > 1e-6).sum())
            total += args.batch
            if not same.all() or (dq[same] > 1e-6).any():
                print(f"seed {seed} {robot} n_goals {n_goals} T {T} grad {grad}: iterations differ in {(itg != ito).sum()}, "
                      f"status in {(stg != sto).sum()}, max|dQ| where equal {dq[same].max():.2e}, overall {dq.max():.2e}", flush=True)
                for b in np.nonzero(dq > 1e-6)[0]:
                    print(f"   instance {b}: iterations {itg[b]} / {ito[b]}, status {stg[b]} / {sto[b]}, f {fg[b]:.12g} / {fo[b]:.12g}, "
                          f"|dQ| {dq[b]:.2e}", flush=True)
            h.close()
    print(f"{total} instances in {time.time() - t0:.0f} s: {bad} off (iteration count, status or > 1e-6 rad), "
          f"max |dQ| over the agreeing ones {worst:.2e}")


if __name__ == "__main__":
    main()
