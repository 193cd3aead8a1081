#!/usr/bin/env python3
"""Where an iteration of k_base_solve (base placement, row f-4) spends its ticks: builds the library with
-DGTO_DEBUG_BASE_TIMING (phase stamps of goal set 0's workgroup, summed over its iterations: gto_kernels.h), solves 1024 goal
sets of ten goals with effort weight 0.01 and 0, and prints ticks per iteration and phase.
usage (GPU box): python tools/base_stamps.py [out.txt]"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = "/tmp/libgto_base_dbg.so"
src = os.path.join(ROOT, "grasptrajopt_amd", "csrc", "gto_api.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-mllvm",
                       "-amdgpu-kernarg-preload-count=16", "-DGTO_DEBUG_BASE_TIMING", f"-I{ROOT}/include", src, "-o", lib], stderr=subprocess.DEVNULL)
os.environ["GTO_HIP_LIB"] = lib
from grasptrajopt_amd import _capi, synthetic as syn  # noqa: E402
from grasptrajopt_amd.robot_desc import load_builtin  # noqa: E402

desc = load_builtin("fetch")
cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", "fetch_cfg.json")))
h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], _capi.default_opts(), device=0, n_gripper_points=100)
qc = np.array(cfg["default_pose"], dtype=np.float64)
sets, ng = 1024, 10
goals, _ = syn.make_base_goal_sets(desc, h.eval_fk, cfg["link_ee"], qc, sets, ng, 0)
QC = np.tile(qc, (sets, 1))
dl = C.CDLL(lib)
names = ["kinematics of the goals (fk_pair_wave)", "goal terms, moments, block entries", "objective, accept / reject", "base gradient, active set",
         "elimination of the goal blocks (8x8 inverses)", "3x3 Schur complement, base step", "back substitution, projection, predicted decrease"]
lines = []
for w in (0.01, 0.0):
    h.solve_base_batch(QC, goals, effort_weight=w)
    buf = (C.c_longlong * 16)()
    dl.gto_debug_base_stamps(buf)  # clear
    y, q, c, it, st = h.solve_base_batch(QC, goals, effort_weight=w)
    dl.gto_debug_base_stamps(buf)
    t = list(buf)
    ev, passes = max(1, t[7]), max(1, t[8])
    tot = sum(t[:7])
    lines.append(f"effort weight {w}: goal set 0: {t[7]} evaluations ({it[0]} iterations), {t[8]} passes of eight goals (half-waves of 32 lanes per goal), "
                 f"{tot / ev:.0f} ticks per evaluation; mean iterations of the call {it.mean():.1f}")
    for i, nme in enumerate(names):
        lines.append(f"   {nme:58s} {t[i] / ev:9.0f} ticks per evaluation  ({100.0 * t[i] / max(tot, 1):4.1f} %)")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
