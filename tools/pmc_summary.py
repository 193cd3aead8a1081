#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV passes (tools/pmc_pass.sh) per kernel: mean counter value per dispatch,
plus calls and mean duration from the kernel trace of the first pass.  Writes pmc_summary.txt and pmc_summary.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
want = ("k_obstacle", "k_lm_step")
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = defaultdict(list)
traces = sorted(glob.glob(os.path.join(out, "*", "**", "*kernel_trace.csv"), recursive=True))
if traces:
    with open(traces[0]) as fh:  # one pass is enough: the profiled arms run at the same clocks
        for row in csv.DictReader(fh):
            dur[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
lines, js = [], {}
for k, cs in sorted(agg.items(), key=lambda kv: -len(kv[1])):
    if not any(w in k for w in want):
        continue
    lines.append(f"## {k}")
    js[k] = {}
    if dur.get(k):
        d = dur[k]
        lines.append(f"{'launches (kernel trace of pass ' + os.path.basename(os.path.dirname(os.path.dirname(traces[0]))) + ')':36s} n={len(d):5d} mean_us={sum(d)/len(d)/1e3:13.2f} max_us={max(d)/1e3:13.2f}")
        js[k]["launches"], js[k]["mean_us"] = len(d), sum(d) / len(d) / 1e3
    for c, v in sorted(cs.items()):
        lines.append(f"{c:36s} n={len(v):5d} mean={sum(v)/len(v):16.1f} max={max(v):16.1f}")
        js[k][c] = sum(v) / len(v)
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "pmc_summary.txt"), "w").write(txt + "\n")
json.dump(js, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
