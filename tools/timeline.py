#!/usr/bin/env python3
"""Per-stream timeline of a rocprofv3 --kernel-trace CSV: for the last `window` ms before the last k_lm_step launch, the
launches of each queue with duration and gap to the previous one.  usage: tools/timeline.py <kernel_trace.csv> [window_ms]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:18], r.get("Queue_Id", "?"), int(r.get("Grid_Size", 0) or 0)))
rows.sort()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
# the timed region of bench.py --steps 20: find the densest stretch with 4 queues active; simply take all rows and report stats per queue
byq = collections.defaultdict(list)
for s, e, n, q, g in rows:
    byq[q].append((s, e, n, g))
print("queues:", {q: len(v) for q, v in byq.items()})
t_end = max(e for s, e, n, q, g in rows if n.startswith("k_lm_step"))
for q, v in sorted(byq.items()):
    v = [x for x in v if t_end - win * 3 <= x[0] <= t_end]
    if len(v) < 10:
        continue
    durs = collections.defaultdict(list)
    gaps = []
    for a, b in zip(v[:-1], v[1:]):
        gaps.append(b[0] - a[1])
    for s, e, n, g in v:
        durs[n].append(e - s)
    print(f"queue {q}: {len(v)} launches in the window; mean gap {sum(gaps)/len(gaps)/1e3:.2f} us; " + "; ".join(f"{n} n={len(d)} mean {sum(d)/len(d)/1e3:.2f} us" for n, d in durs.items()))
# overlap: how many kernels are running at once in the window
ev = []
for s, e, n, q, g in rows:
    if t_end - win * 3 <= s <= t_end:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("kernels in flight (share of the window):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
