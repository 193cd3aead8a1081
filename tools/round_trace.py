#!/usr/bin/env python3
"""Per-round listing of solver calls from a rocprofv3 --kernel-trace CSV (rounds mode): for the LAST group of calls that
overlap in time (the timed region of `bench.py --steps 20`), every round of every queue: start offset, obstacle kernel
duration, gap, step kernel duration, gap to the next round (us).
usage: [MIN_CALLS=1] tools/round_trace.py <kernel_trace.csv> [call index from the end, default 0]   (MIN_CALLS=1: calls that run alone count as groups)"""
import collections
import csv
import os
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"),
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
rows.sort()
byq = collections.defaultdict(list)
for row in rows:
    byq[row[3]].append(row)
calls = []
for q, v in byq.items():
    cur = None
    for s, e, n, _, g, w in v:
        if n == "k_lm_init":
            cur = {"q": q, "t0": s, "ev": [], "B": g // max(w, 1)}
        elif cur is not None and n in ("k_obstacle_gram", "k_lm_step", "k_lm_step_wide"):
            cur["ev"].append((s, e, "k_lm_step" if n == "k_lm_step_wide" else n, g // max(w, 1)))
        elif cur is not None and n == "k_lm_finalize":
            cur["t1"] = e
            calls.append(cur)
            cur = None
calls.sort(key=lambda c: c["t0"])
# groups of calls that overlap in time
groups, cur = [], []
for c in calls:
    if cur and c["t0"] > max(x["t1"] for x in cur):
        groups.append(cur)
        cur = []
    cur.append(c)
if cur:
    groups.append(cur)
groups = [g for g in groups if len(g) >= int(os.environ.get("MIN_CALLS", "2")) and all(len(c["ev"]) >= 20 for c in g)]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = groups[-1 - which]
t0 = min(c["t0"] for c in g)
print(f"{len(groups)} groups of overlapping calls; group -{1+which}: {len(g)} calls of B = {[c['B'] for c in g]}, span {(max(c['t1'] for c in g) - t0)/1e3:.1f} us")
for ci, c in enumerate(g):
    ev = c["ev"]
    print(f"-- call {ci} queue {c['q']}: {len(ev)//2} rounds, {(c['t1']-c['t0'])/1e3:.1f} us")
    i, k = 0, 0
    while i + 1 < len(ev):
        if ev[i][2] == "k_obstacle_gram" and ev[i + 1][2] == "k_lm_step":
            o, s = ev[i], ev[i + 1]
            nxt = ev[i + 2][0] if i + 2 < len(ev) else s[1]
            print(f"  r{k:3d} @{(o[0]-t0)/1e3:8.1f}  obs {(o[1]-o[0])/1e3:6.1f} (grid {o[3]:5d})  gap {(s[0]-o[1])/1e3:5.1f}  step {(s[1]-s[0])/1e3:5.1f} (grid {s[3]:4d}, {'8w' if False else ''})  gap {(nxt-s[1])/1e3:5.1f}  round {(nxt-o[0])/1e3:6.1f}")
            i += 2
            k += 1
        else:
            i += 1
