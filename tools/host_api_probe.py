#!/usr/bin/env python3
"""Per-call wall time of the host-pointer entry point (gto_solve_batch) when LANES threads call it at once, each on its own
handle and stream with calls of CALL instances (the driver's 20-step region: 4 lanes x 320): distribution over REPS rounds,
next to the same calls through the device-resident entry point.  usage (through gpurun): python tools/host_api_probe.py"""
import json, os, sys, time, threading
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from grasptrajopt_amd import _capi, synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "panda_cfg.json")))
desc = load_builtin("panda_5k")
opts = _capi.default_opts()
LANES, CALL, REPS = int(os.environ.get("LANES", "4")), int(os.environ.get("CALL", "320")), int(os.environ.get("REPS", "12"))
dev = torch.device("cuda", 0)
sc = syn.make_scene(0, n=128, res=0.0175)
hs, args, dargs = [], [], []
for d in range(LANES):
    h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
    st = torch.cuda.Stream(dev)
    h.set_stream(st.cuda_stream)
    if d == 0:
        h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    else:
        h.share_scene(0, hs[0][0])
    RT, qg = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], CALL, seed=3 + 17 * d)
    qc = np.tile(np.array(cfg["default_pose"]), (CALL, 1))
    Q0 = np.stack([syn.make_seed(qc[i], qg[i], 50, desc.param_index) for i in range(CALL)])
    S = np.tile(syn.standoff_pose(-0.1, "z").reshape(1, 16), (CALL, 1))
    a = (np.zeros(CALL, np.int32), qc, RT.reshape(CALL, 1, 16), np.ones(CALL, np.int32), S, np.zeros((CALL, 3)), Q0)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(dt).to(dev)
    inp = [t(a[0], torch.int32), t(a[1], torch.float64), t(a[2], torch.float64), t(a[3], torch.int32), t(a[4], torch.float64), t(a[5], torch.float64), t(a[6], torch.float64)]
    out = [torch.empty((CALL, desc.ndof, 50), dtype=torch.float64, device=dev), torch.empty((CALL, desc.ndof, 49), dtype=torch.float64, device=dev),
           torch.empty(CALL, dtype=torch.float64, device=dev), torch.empty(CALL, dtype=torch.int32, device=dev), torch.empty(CALL, dtype=torch.int32, device=dev)]
    hs.append((h, st)); args.append(a); dargs.append(inp + out)
torch.cuda.synchronize(dev)

def run(kind):
    res = np.zeros((REPS, LANES)); wall = np.zeros(REPS)
    bar = threading.Barrier(LANES + 1)
    def worker(d):
        h, st = hs[d]
        for r in range(REPS + 2):
            bar.wait()
            t0 = time.perf_counter()
            if kind == "host":
                h.solve_batch(*args[d])
            else:
                h.solve_batch_device(CALL, 1, *[x.data_ptr() for x in dargs[d]], st.cuda_stream)
                st.synchronize()
            if r >= 2:
                res[r - 2, d] = time.perf_counter() - t0
            bar.wait()
    th = [threading.Thread(target=worker, args=(d,)) for d in range(LANES)]
    [t.start() for t in th]
    for r in range(REPS + 2):
        bar.wait(); t0 = time.perf_counter(); bar.wait()
        if r >= 2:
            wall[r - 2] = time.perf_counter() - t0
    [t.join() for t in th]
    return res * 1e3, wall * 1e3

for kind in ("device", "host", "device", "host"):
    res, wall = run(kind)
    print(f"{kind:6s}: round wall ms: median {np.median(wall):.2f} min {wall.min():.2f} max {wall.max():.2f} | all {np.round(wall, 2).tolist()}")
    print(f"        per-lane call ms median {np.round(np.median(res, axis=0), 2).tolist()}  -> {LANES*CALL/np.median(wall)*1e3:.0f} traj/s at the median")
for h, _ in reversed(hs):
    h.close()
