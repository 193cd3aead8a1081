#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV passes (tools/pmc_pass.sh) per kernel: mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for k, cs in sorted(agg.items(), key=lambda kv: -len(kv[1])):
    if not (k.startswith("k_obstacle") or k.startswith("k_lm_step") or "k_traj_solve" in k):
        continue
    lines.append(f"## {k}")
    for c, v in sorted(cs.items()):
        lines.append(f"{c:36s} n={len(v):5d} mean={sum(v)/len(v):16.1f} max={max(v):16.1f}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "pmc_summary.txt"), "w").write(txt + "\n")
