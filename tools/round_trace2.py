#!/usr/bin/env python3
"""Rounds of the LAST solver call in a rocprofv3 --kernel-trace CSV: per round obstacle / step durations and grids."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["LDS_Block_Size"])))
rows.sort()
last_init = max(i for i, r in enumerate(rows) if r[2] == "k_lm_init")
ev = rows[last_init:]
t0 = ev[0][0]
k = 0
tot_o = tot_s = 0
for i, (s, e, n, g, lds) in enumerate(ev):
    if n == "k_obstacle_gram" and i + 1 < len(ev) and ev[i + 1][2] == "k_lm_step":
        s2, e2, _, g2, lds2 = ev[i + 1]
        nxt = ev[i + 2][0] if i + 2 < len(ev) else e2
        print(f"r{k:3d} @{(s-t0)/1e3:8.1f} obs {(e-s)/1e3:6.1f} (grid {g:5d}) gap {(s2-e)/1e3:4.1f} step {(e2-s2)/1e3:5.1f} (grid {g2:4d}, lds {lds2}) gap {(nxt-e2)/1e3:4.1f} round {(nxt-s)/1e3:6.1f}")
        k += 1; tot_o += e - s; tot_s += e2 - s2
print(f"total {(ev[-1][1]-t0)/1e3:.1f} us, {k} rounds, obstacle {tot_o/1e3:.1f}, step {tot_s/1e3:.1f}")
