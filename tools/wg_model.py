#!/usr/bin/env python3
"""What a workgroup of the obstacle kernel costs by the number of chunks that survive its broad phase: builds the library with
-DGTO_DEBUG_LONGEST_WG (per workgroup: surviving chunks and clock64 ticks from entry to exit, summed per chunk count:
gto_kernels.h), runs one lane of a bench workload under GTO_DEBUG_TIMING and fits ticks = fixed + per_chunk x chunks over the
workgroups of the last call.  The ticks are those of a workgroup that shares its SIMDs with four others: a split of where a
launch's time goes, not an isolated latency.
usage (GPU box): python tools/wg_model.py <out.txt> [bench.py arguments of the workload; default: BASELINE configs[2]]"""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = "/tmp/libgto_wg_dbg.so"
src = os.path.join(ROOT, "grasptrajopt_amd", "csrc", "gto_api.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-mllvm",
                       "-amdgpu-kernarg-preload-count=16", "-DGTO_DEBUG_LONGEST_WG", f"-I{ROOT}/include", src, "-o", lib], stderr=subprocess.DEVNULL)
out = sys.argv[1]
argv = sys.argv[2:] or ["--robot", "fetch", "--batch", "256", "--shelf", "--merge", "8"]
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--light", "--no-cpu-baseline", "--no-next-rows", "--no-other-configs", "--repeats", "1",
       "--warmup", "1", "--steps", "4", "--pipeline", "1"] + argv
r = subprocess.run(cmd, env=dict(os.environ, GTO_DEBUG_TIMING="1", GTO_HIP_LIB=lib), capture_output=True, text=True, timeout=900)
lines = r.stderr.splitlines()


def cells(tag):
    l = [x for x in lines if tag in x][-1]
    return np.array([int(v) for v in l.split("):")[-1].split()], dtype=np.float64)


n, t = cells("surviving chunks per workgroup"), cells("ticks those workgroups ran")
longest = [x for x in lines if "longest regular obstacle workgroup" in x][-1]
k = np.arange(64)
sel = (n > 0) & (k > 0)
A = np.stack([np.ones(sel.sum()), k[sel]], 1) * np.sqrt(n[sel])[:, None]
fixed, per_chunk = np.linalg.lstsq(A, (t[sel] / n[sel]) * np.sqrt(n[sel]), rcond=None)[0]
tot = t.sum()
rows = [f"# workload: bench.py {' '.join(argv)} (one lane, last call of the run); library built with -DGTO_DEBUG_LONGEST_WG",
        longest.replace("[gto dbg] ", "# "),
        f"workgroups {int(n.sum())}, of them without a surviving chunk {int(n[0])} ({n[0] / n.sum():.3f}), mean surviving chunks of the others {(k * n)[1:].sum() / n[1:].sum():.1f}",
        f"fit over the workgroups with survivors: ticks = {fixed:.0f} + {per_chunk:.0f} x chunks",
        f"share of all workgroup ticks: workgroups without survivors {t[0] / tot:.3f} | fixed part of the others {fixed * n[1:].sum() / tot:.3f} | per-chunk part {per_chunk * (k * n)[1:].sum() / tot:.3f}",
        "chunks  workgroups  mean ticks"]
for i in range(64):
    if n[i] > 0:
        rows.append(f"{i:6d}  {int(n[i]):10d}  {t[i] / n[i]:10.0f}")
open(out, "w").write("\n".join(rows) + "\n")
print("\n".join(rows[:6]))
