#!/usr/bin/env python3
"""many_scenes.py — BASELINE configs[3] on one GPU: S synthetic scenes x 8 grasps each (every instance of a call
reads a different 128^3 field: 148 MB resident per scene, no reuse between instances in L2).
Usage: python tools/many_scenes.py [--scenes 64] [--grasps 8] [--lanes 4] [--calls 3]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=64)
    ap.add_argument("--grasps", type=int, default=8)
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--calls", type=int, default=3)
    args = ap.parse_args()
    import torch
    from grasptrajopt_amd import _capi, synthetic as syn
    from grasptrajopt_amd.robot_desc import load_builtin

    dev = torch.device("cuda", 0)
    cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
    desc = load_builtin("panda_5k")
    opts = _capi.default_opts()
    T, ndof = opts.T, desc.ndof
    S, G = args.scenes, args.grasps
    B = S * G
    lanes = []
    for i in range(args.lanes):
        h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
        st = torch.cuda.Stream(dev)
        h.set_stream(st.cuda_stream)
        lanes.append((h, st))
    h0 = lanes[0][0]
    moving = desc.link_is_moving()[desc.point_link]
    t0 = time.perf_counter()
    RT, qg, sid = [], [], []
    for s in range(S):
        sc = syn.make_scene(s, n=128, res=2.24 / 128)
        h0.set_scene(s, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
        for h, _ in lanes[1:]:
            h.share_scene(s, h0)

        def cc(q, s=s):
            _, _, val, _ = h0.eval_points(s, q, [0.0, 0.0, 0.0], use_obs=True)
            return (val * moving[None, :]).sum(axis=1)

        r, q = syn.make_goals(desc, h0.eval_fk, cfg["link_ee"], G, seed=s, collision_cost=cc)
        RT.append(r)
        qg.append(q)
        sid += [s] * G
    RT, qg = np.concatenate(RT), np.concatenate(qg)
    print(f"{S} scenes resident ({S * 0.148:.1f} GB) and {B} goal grasps in {time.perf_counter() - t0:.1f} s", flush=True)
    qc = np.tile(np.array(cfg["default_pose"]), (B, 1))
    Q0 = np.stack([syn.make_seed(qc[b], qg[b], T, desc.param_index) for b in range(B)])
    Sd = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (B, 1))
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    jobs = []
    for h, st in lanes:
        bufs = [t(np.array(sid), torch.int32), t(qc, torch.float64), t(RT.reshape(B, 1, 16), torch.float64),
                torch.ones(B, dtype=torch.int32, device=dev), t(Sd, torch.float64), t(np.zeros((B, 3)), torch.float64),
                t(Q0, torch.float64), torch.empty((B, ndof, T), dtype=torch.float64, device=dev),
                torch.empty((B, ndof, T - 1), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.float64, device=dev),
                torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)]
        jobs.append((h, st, bufs))
    torch.cuda.synchronize()

    def run(job, n):
        h, st, bufs = job
        for _ in range(n):
            h.solve_batch_device(B, 1, *[x.data_ptr() for x in bufs], st.cuda_stream)
        st.synchronize()

    for j in jobs:
        run(j, 1)
    th = [threading.Thread(target=run, args=(j, args.calls)) for j in jobs]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    it = jobs[0][2][10].cpu().numpy()
    same = all(bool(torch.equal(j[2][7], jobs[0][2][7])) for j in jobs[1:])
    print(f"{args.lanes} lanes x {args.calls} calls of {B} instances ({S} scenes x {G} grasps): {B * args.calls * args.lanes / el:.0f} trajectories/s, "
          f"iterations mean {it.mean():.1f} max {it.max()}, lanes bit-identical {same}")
    for h, _ in reversed(lanes):
        h.close()


if __name__ == "__main__":
    main()
