#!/usr/bin/env python3
"""Round timeline of solver calls from a rocprofv3 --kernel-trace CSV (rounds mode): a call on a queue is the stretch
from k_lm_init to k_lm_finalize; inside it every k_obstacle_gram + k_lm_step pair is one round.  Prints, for the calls
that ran while at least `min_queues` queues were busy (the pipelined phase of bench.py), the duration of a call, the
number of rounds, and how the round time (obstacle kernel, step kernel, gap) develops along the call in tenths of the
round count; plus the share of the time with 0..n kernels in flight.
usage: tools/timeline.py <kernel_trace.csv> [min_queues]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
min_q = int(sys.argv[2]) if len(sys.argv) > 2 else 3
byq = collections.defaultdict(list)
for s, e, n, q in rows:
    byq[q].append((s, e, n))
calls = []  # (queue, t0, t1, rounds[(obs_dur, step_dur, round_span)])
for q, v in byq.items():
    cur = None
    for s, e, n in v:
        if n == "k_lm_init":
            cur = {"q": q, "t0": s, "ev": []}
        elif cur is not None and n in ("k_obstacle_gram", "k_lm_step"):
            cur["ev"].append((s, e, n))
        elif cur is not None and n == "k_lm_finalize":
            cur["t1"] = e
            calls.append(cur)
            cur = None
if not calls:
    raise SystemExit("no k_lm_init .. k_lm_finalize stretch found")


def busy_queues(t0, t1):
    return sum(1 for c in calls if c["t0"] < t1 and c["t1"] > t0)


sel = [c for c in calls if busy_queues(c["t0"], c["t1"]) >= min_q and len(c["ev"]) >= 40]
if not sel:
    sel = [c for c in calls if len(c["ev"]) >= 40]
sizes = collections.Counter(len(c["ev"]) // 2 for c in sel)
print(f"{len(calls)} calls, {len(sel)} of them with >= {min_q} queues busy; rounds per call: min {min(sizes)} max {max(sizes)}")
dur = [(c["t1"] - c["t0"]) / 1e6 for c in sel]
print(f"call duration ms: mean {sum(dur)/len(dur):.2f} min {min(dur):.2f} max {max(dur):.2f}")
# rounds: obstacle launch followed by step launch
dec = [collections.defaultdict(list) for _ in range(10)]
for c in sel:
    ev = c["ev"]
    rounds = []
    i = 0
    while i + 1 < len(ev):
        if ev[i][2] == "k_obstacle_gram" and ev[i + 1][2] == "k_lm_step":
            nxt = ev[i + 2][0] if i + 2 < len(ev) else ev[i + 1][1]
            rounds.append((ev[i][1] - ev[i][0], ev[i + 1][1] - ev[i + 1][0], nxt - ev[i][0]))
            i += 2
        else:
            i += 1
    n = len(rounds)
    for k, (o, s, span) in enumerate(rounds):
        d = dec[min(9, 10 * k // n)]
        d["obs"].append(o); d["step"].append(s); d["span"].append(span)
print("tenth of the call | rounds | round span us | obstacle us | step us | share of the call time")
tot = sum(sum(d["span"]) for d in dec)
for k, d in enumerate(dec):
    if d["span"]:
        m = lambda x: sum(x) / len(x) / 1e3
        print(f"  {k}  {len(d['span']):6d}  {m(d['span']):8.1f}  {m(d['obs']):8.1f}  {m(d['step']):8.1f}  {sum(d['span'])/tot:6.3f}")
# kernels in flight over the selected calls' time range
t0, t1 = min(c["t0"] for c in sel), max(c["t1"] for c in sel)
ev = []
for s, e, n, q in rows:
    if s >= t0 and e <= t1:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tt = sum(hist.values())
print("kernels in flight (share of the time):", {k: round(v / tt, 3) for k, v in sorted(hist.items())})
