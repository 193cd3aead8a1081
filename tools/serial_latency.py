#!/usr/bin/env python3
"""Latency of one solver call at a time (rounds mode): B instances of the bench workload (Panda-5k, 128^3), device-resident
entry point, median over REPS calls; with FORCE_ITERS=1 the stopping tolerances are switched off so that every instance runs
max_iter iterations (the chain of a straggler: time / max_iter = one round).  Env knobs are read at handle creation.
usage (through gpurun): [B=1,4,64] [REPS=20] [FORCE_ITERS=1] python tools/serial_latency.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from grasptrajopt_amd import _capi, synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
desc = load_builtin("panda_5k")
opts = _capi.default_opts()
force = os.environ.get("FORCE_ITERS", "0") != "0"
if force:
    opts.tol_rel_f, opts.tol_step = 0.0, 0.0
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
sc = syn.make_scene(0, n=128, res=0.0175)
h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
h.set_stream(stream.cuda_stream)
reps = int(os.environ.get("REPS", "20"))
for B in [int(x) for x in os.environ.get("B", "1,4,64").split(",")]:
    RT, qg = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=3)
    qc = np.tile(np.array(cfg["default_pose"]), (B, 1))
    Q0 = np.stack([syn.make_seed(qc[i], qg[i], 50, desc.param_index) for i in range(B)])
    S = np.tile(syn.standoff_pose(-0.1, "z").reshape(1, 16), (B, 1))
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    inp = [torch.zeros(B, dtype=torch.int32, device=dev), t(qc, torch.float64), t(RT.reshape(B, 1, 16), torch.float64),
           torch.ones(B, dtype=torch.int32, device=dev), t(S, torch.float64), torch.zeros((B, 3), dtype=torch.float64, device=dev), t(Q0, torch.float64)]
    out = [torch.empty((B, desc.ndof, 50), dtype=torch.float64, device=dev), torch.empty((B, desc.ndof, 49), dtype=torch.float64, device=dev),
           torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)]
    ptrs = [x.data_ptr() for x in inp + out]
    ts = []
    for r in range(reps + 3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        h.solve_batch_device(B, 1, *ptrs, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[3:]) * 1e3
    it = out[3].cpu().numpy()
    print(f"B={B:4d} force={int(force)}: call ms median {np.median(ts):.3f} min {ts.min():.3f} max {ts.max():.3f} | iters mean {it.mean():.1f} max {it.max()} | "
          f"us per iteration of the longest chain {1e3*np.median(ts)/max(it.max(),1):.1f}", flush=True)
h.close()
