#!/usr/bin/env python3
"""Lanes of ONE solve call from a rocprofv3 --kernel-trace CSV: for the last call in the trace (k_lm_init ... k_lm_finalize
on the caller's queue), per hardware queue the rounds it ran (obstacle + step launches), their durations and gaps, the
hand-overs (k_adopt), and a coarse time line (launches and busy time per queue per bucket).
usage: tools/lane_trace.py <kernel_trace.csv> [bucket_us=250] [call index from the end=0]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"),
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))))
rows.sort()
bucket = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fins = [i for i, r in enumerate(rows) if r[2].startswith("k_lm_finalize")]
calls = []  # (finalize index, init indices): the biggest calls of the trace (the timed regions), `back` from the end
for j, f in enumerate(fins):
    calls.append((f, [i for i in range(fins[j - 1] + 1 if j else 0, f) if rows[i][2].startswith("k_lm_init")]))
size = lambda c: sum(rows[i][4] for i in c[1])
big = max(size(c) for c in calls)
fi, inits = [c for c in calls if size(c) == big][-1 - back]
t0, t1 = rows[inits[0]][0], rows[fi][1]
ev = [r for r in rows[inits[0]:fi + 1]]
print(f"call: {(t1 - t0) / 1e3:.1f} us from the first k_lm_init to the end of k_lm_finalize, {len(ev)} launches, instances {sum(rows[i][4] for i in inits)}")
byq = collections.defaultdict(list)
for r in ev:
    byq[r[3]].append(r)
for q, v in sorted(byq.items()):
    obs = [r for r in v if r[2].startswith("k_obstacle_gram")]
    stp = [r for r in v if r[2].startswith("k_lm_step")]
    ad = [r for r in v if r[2].startswith("k_adopt")]
    if not stp:
        print(f"queue {q}: {[r[2][:24] for r in v][:6]}")
        continue
    busy = sum(r[1] - r[0] for r in v) / 1e3
    span = (v[-1][1] - v[0][0]) / 1e3
    print(f"queue {q}: rounds {len(stp)}  first launch at {(v[0][0] - t0) / 1e3:8.1f} us  last end at {(v[-1][1] - t0) / 1e3:8.1f} us  busy {busy:8.1f} us of {span:8.1f}"
          f"  obs mean {sum(r[1] - r[0] for r in obs) / max(1, len(obs)) / 1e3:6.1f} us  step mean {sum(r[1] - r[0] for r in stp) / len(stp) / 1e3:6.1f} us"
          f"  adopts at {[round((r[0] - t0) / 1e3) for r in ad]}")
nb = int((t1 - t0) / 1e3 / bucket) + 1
print(f"time line, buckets of {bucket:.0f} us: per queue  launches / busy us / mean step grid")
for b in range(nb):
    lo, hi = t0 + b * bucket * 1e3, t0 + (b + 1) * bucket * 1e3
    cells = []
    for q, v in sorted(byq.items()):
        inb = [r for r in v if lo <= r[0] < hi]
        st = [r[4] for r in inb if r[2].startswith("k_lm_step")]
        cells.append(f"{len(inb):3d}/{sum(min(r[1], hi) - r[0] for r in inb) / 1e3:5.0f}/{(sum(st) // len(st)) if st else 0:4d}")
    print(f"{b * bucket:7.0f}  " + "   ".join(cells))
