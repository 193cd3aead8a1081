#!/usr/bin/env python3
"""Lanes of ONE solve call from a rocprofv3 --kernel-trace CSV (gto_set_lanes: a call's instances dealt to several streams):
for the last call of the trace that finalised <instances> instances, per hardware queue the rounds it ran (obstacle + step
launches), mean durations, the hand-overs to lane 0 (k_adopt), and a coarse time line: per bucket and queue
launches / busy us / mean step grid (= instances in flight).
usage: tools/lane_trace.py <kernel_trace.csv> <instances of the call> [bucket_us=250]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"),
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))))
rows.sort()
total = int(sys.argv[2])
bucket = float(sys.argv[3]) if len(sys.argv) > 3 else 250.0
fins = [i for i, r in enumerate(rows) if r[2].startswith("k_lm_finalize") and r[4] == total]
fi = fins[-1]
j = fi
while not rows[j][2].startswith("k_lm_init"):
    j -= 1
while rows[j - 1][2].startswith("k_lm_init"):
    j -= 1
t0, t1 = rows[j][0], rows[fi][1]
ev = rows[j:fi + 1]
print(f"call of {total} instances: {(t1 - t0) / 1e3:.1f} us from the first k_lm_init to the end of k_lm_finalize, {len(ev)} launches (under the tracer)")
byq = collections.defaultdict(list)
for r in ev:
    byq[r[3]].append(r)
for q, v in sorted(byq.items()):
    obs = [r for r in v if r[2].startswith("k_obstacle")]
    stp = [r for r in v if r[2].startswith("k_lm_step")]
    ad = [r for r in v if r[2].startswith("k_adopt")]
    if not stp:
        print(f"  queue {q}: {[r[2][:22] for r in v][:8]}")
        continue
    full_o = [r[1] - r[0] for r in obs if r[4] >= 0.9 * max(x[4] for x in obs)]
    print(f"  queue {q}: {len(stp)} rounds, first launch at {(v[0][0] - t0) / 1e3:.0f} us, last end at {(v[-1][1] - t0) / 1e3:.0f} us, busy {sum(r[1] - r[0] for r in v) / 1e3:.0f} us; "
          f"obstacle launch mean {sum(r[1] - r[0] for r in obs) / len(obs) / 1e3:.1f} us (full-size launches {sum(full_o) / max(1, len(full_o)) / 1e3:.1f}), "
          f"step launch mean {sum(r[1] - r[0] for r in stp) / len(stp) / 1e3:.1f} us; hand-overs taken at {[round((r[0] - t0) / 1e3) for r in ad]} us")
print(f"time line, buckets of {bucket:.0f} us, per queue: launches / busy us / mean step grid")
for b in range(int((t1 - t0) / 1e3 / bucket) + 1):
    lo, hi = t0 + b * bucket * 1e3, t0 + (b + 1) * bucket * 1e3
    cells = []
    for q, v in sorted(byq.items()):
        inb = [r for r in v if lo <= r[0] < hi]
        st = [r[4] for r in inb if r[2].startswith("k_lm_step")]
        cells.append(f"{len(inb):3d}/{sum(min(r[1], hi) - r[0] for r in inb) / 1e3:4.0f}/{(sum(st) // len(st)) if st else 0:4d}")
    print(f"  {b * bucket:6.0f}  " + "   ".join(cells))
