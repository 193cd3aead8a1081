#!/usr/bin/env python3
"""pipeline_probe.py — how much of the GPU does ONE batch of 64 solves use?

(a) batch-size sweep through one handle; (b) N handles on N streams, one host thread each, every
thread solving `steps` batches of 64 back to back (ctypes releases the GIL during the call).
Prints one line per configuration.  Usage: python tools/pipeline_probe.py [--steps 10]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batches", default="64,128,256,512")
    ap.add_argument("--handles", default="1,2,3,4")
    args = ap.parse_args()
    import torch
    from grasptrajopt_amd import _capi, synthetic as syn
    from grasptrajopt_amd.robot_desc import load_builtin

    dev = torch.device("cuda", 0)
    cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
    desc = load_builtin("panda_5k")
    opts = _capi.default_opts()
    T, ndof = opts.T, desc.ndof
    sc = syn.make_scene(0, n=128, res=2.24 / 128)
    moving = desc.link_is_moving()[desc.point_link]

    class Job:
        def __init__(self, B, seed):
            self.B = B
            self.h = h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
            h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)

            def cc(q):
                _, _, val, _ = h.eval_points(0, q, [0.0, 0.0, 0.0], use_obs=True)
                return (val * moving[None, :]).sum(axis=1)

            RT, qg = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=seed, collision_cost=cc)
            qc = np.tile(np.array(cfg["default_pose"]), (B, 1))
            Q0 = np.stack([syn.make_seed(qc[b], qg[b], T, desc.param_index) for b in range(B)])
            S = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (B, 1))
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
            self.bufs = [torch.zeros(B, dtype=torch.int32, device=dev), t(qc, torch.float64), t(RT.reshape(B, 1, 16), torch.float64),
                         torch.ones(B, dtype=torch.int32, device=dev), t(S, torch.float64), t(np.zeros((B, 3)), torch.float64),
                         t(Q0, torch.float64), torch.empty((B, ndof, T), dtype=torch.float64, device=dev),
                         torch.empty((B, ndof, T - 1), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.float64, device=dev),
                         torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)]
            self.stream = torch.cuda.Stream(dev)

        def step(self):
            self.h.solve_batch_device(self.B, 1, *[b.data_ptr() for b in self.bufs], self.stream.cuda_stream)

    for B in [int(x) for x in args.batches.split(",")]:
        j = Job(B, 0)
        for _ in range(2):
            j.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            j.step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        it = j.bufs[10].cpu().numpy()
        print(f"batch {B:5d}: {1e3 * el / args.steps:8.3f} ms/step  {B * args.steps / el:10.1f} traj/s  iters mean {it.mean():.1f} max {it.max()}",
              flush=True)
        j.h.close()

    for N in [int(x) for x in args.handles.split(",")]:
        jobs = [Job(64, 0) for _ in range(N)]
        for j in jobs:
            j.step()
        torch.cuda.synchronize()
        bar = threading.Barrier(N + 1)

        def run(j):
            bar.wait()
            for _ in range(args.steps):
                j.step()
            j.stream.synchronize()

        th = [threading.Thread(target=run, args=(j,)) for j in jobs]
        for x in th:
            x.start()
        bar.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"handles {N}: {1e3 * el / args.steps:8.3f} ms per {N} step(s)  {64 * N * args.steps / el:10.1f} traj/s", flush=True)
        for j in jobs:
            j.h.close()


if __name__ == "__main__":
    main()
