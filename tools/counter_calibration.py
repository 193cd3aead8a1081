#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE against known byte counts (tools/probes/gather32_probe.hip): runs the probe plain (its own
timings) and under rocprofv3 --pmc in separate passes (FETCH_SIZE, WRITE_SIZE, the L2's memory-side request counters),
prints per kernel: bytes by construction, the counters per launch (KB -> bytes), and the factor that turns a counter into
the bytes the kernel moved.  On the GPU box:  python tools/counter_calibration.py gpurun_out/calib  > gpurun_out/calib.txt"""
import csv
import glob
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "calib"))
os.makedirs(out, exist_ok=True)
exe = "/tmp/gather32_probe"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "probes", "gather32_probe.hip"), "-o", exe])
plain = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
known = {}
for line in plain.splitlines():
    m = re.match(r"PROBE (.+?)\s+bytes_by_construction (\d+)\s+best_ms ([\d.]+)\s+GB/s_by_construction ([\d.]+).*# (.*)", line)
    if m:
        known[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5))
passes = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "ea_rd": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
          "ea_wr": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"], "tcc": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"]}
env = dict(os.environ, TMPDIR="/tmp")
agg = defaultdict(lambda: defaultdict(list))
for name, ctrs in passes.items():
    d = os.path.join(out, name)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "p", "--", exe], cwd="/tmp", env=env,
                       capture_output=True, text=True)
    if r.returncode != 0:
        print(f"# pass {name} ({' '.join(ctrs)}) failed: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else r.returncode}")
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# rocprofv3 counters against known byte counts (tools/probes/gather32_probe.hip; 4 launches per kernel, means per launch)")
print("# FETCH_SIZE / WRITE_SIZE are reported in KB (x 1024 below).  factor = bytes by construction / counter bytes")
for k, (b, ms, gbs, what) in known.items():
    c = {n: sum(v) / len(v) for n, v in agg.get(k, {}).items()}
    print(f"## {k}: {what}")
    print(f"   bytes by construction {b:.4g}   best launch {ms:.4f} ms = {gbs:.1f} GB/s of useful bytes")
    if "FETCH_SIZE" in c:
        fb = c["FETCH_SIZE"] * 1024
        print(f"   FETCH_SIZE {fb:.4g} B per launch   useful/FETCH_SIZE = {b / max(fb, 1):.3f}   (FETCH_SIZE x 2 = {2 * fb:.4g}: {2 * fb / b:.3f} x useful)")
    if "WRITE_SIZE" in c:
        wb = c["WRITE_SIZE"] * 1024
        print(f"   WRITE_SIZE {wb:.4g} B per launch   useful/WRITE_SIZE = {b / max(wb, 1):.3f}")
    for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
        if n in c:
            print(f"   {n} {c[n]:.4g} per launch   useful bytes per request = {b / max(c[n], 1):.1f}")
