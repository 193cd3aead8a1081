#!/usr/bin/env python3
"""Critical path of ONE workgroup of the step kernels from the kernel's own stamps (GTO_DEBUG_TIMING: clock64 ticks of
instance 0's workgroup, printed by gto_solve_batch_device at the end of a call): k_lm_step<4,1> in a call that fills the GPU
(320 instances, the launch that carries the broad phase of the next job in its tail) and k_lm_step<8,4> with one instance.
Writes profiles/<tag>_step_stamps.json; tools/traffic_json.py puts critical_path_cycles / (launch duration x ticks per us) next
to the PMC figures of the same variant.   usage (GPU box): python tools/step_stamps.py <tag>"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["GTO_ROOT"])
from grasptrajopt_amd import _capi, synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin
B, max_iter = int(sys.argv[1]), int(sys.argv[2])
cfg = json.load(open(os.path.join(os.environ["GTO_ROOT"], "grasptrajopt_amd", "data", "panda_cfg.json")))
desc = load_builtin("panda_5k")
opts = _capi.default_opts(); opts.max_iter = max_iter
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
sc = syn.make_scene(0, n=128, res=2.24 / 128, origin=(-0.4, -1.12, -0.4), table_z=-0.03)
h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
RT, qg = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=0, zlim=(0.08, 0.7))
qc = np.tile(np.array(cfg["default_pose"], dtype=np.float64), (B, 1))
Q0 = np.stack([syn.make_seed(qc[b], qg[b], opts.T, desc.param_index) for b in range(B)])
S = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (B, 1))
for _ in range(3):
    h.solve_batch(0, qc, RT.reshape(B, 1, 16), 1, S, np.zeros((B, 3)), Q0)
'''


def run(B, max_iter):
    env = dict(os.environ, GTO_DEBUG_TIMING="1", GTO_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", WORKER, str(B), str(max_iter)], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stderr.splitlines() if l.startswith("[gto dbg]")]
    ph = [l for l in lines if "step-kernel phases" in l][-1]
    tl = [l for l in lines if "broad phase in the step kernel's tail" in l][-1]
    nums = lambda l: [int(x) for x in re.findall(r"(?<= )(-?\d+)(?= \||$)", l.split("(cycles)", 1)[1])]
    return ph, tl, nums(ph), nums(tl), r.stderr[-400:] if r.returncode else ""


tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = {"what": "clock64 ticks of instance 0's workgroup (GTO_DEBUG_TIMING), last launch of the call that wrote them; ticks_per_us: the "
               "shader clock the counter runs at (2.4 GHz; GRBM_GUI_ACTIVE over the launch duration gives 2.4-2.5)", "variants": {}}
ph, tl, a, b, err = run(320, 6)
steps, tail = sum(a[:7]), sum(x for x in b if x > 0)  # (the P5 stamp is taken behind the tail: steps includes it)
out["variants"]["k_lm_step<4,1>"] = {"critical_path_cycles": steps, "ticks_per_us": 2400, "step_phases": ph, "tail_phases": tl,
                                     "cycles_tail_broad_phase": tail, "cycles_without_tail": steps - tail}
ph, tl, a, b, err2 = run(1, 100)
out["variants"]["k_lm_step<8,4>"] = {"critical_path_cycles": sum(a[:7]), "ticks_per_us": 2400, "step_phases": ph}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_step_stamps.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
