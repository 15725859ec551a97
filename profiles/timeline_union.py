#!/usr/bin/env python3
"""Where a step of overlapping chains goes, from a rocprofv3 kernel trace: over the last `frac` of the trace the wall time, the
time at least one kernel was running (union of the intervals: chains overlap, so the sum of durations overstates it), the idle
rest, and per kernel name the launches, the summed duration and the time it ran ALONE (no other kernel in flight).
    python profiles/timeline_union.py <kernel_trace.csv> [frac=0.5] [steps in that part]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = rows[int(len(rows) * (1 - frac)):]
name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("c2r::", "")
ev = []
for i, r in enumerate(rows):
    ev.append((int(r["Start_Timestamp"]), 1, i)); ev.append((int(r["End_Timestamp"]), 0, i))
ev.sort()
live, last, union, alone = set(), None, 0, collections.Counter()
for t, kind, i in ev:
    if live:
        union += t - last
        if len(live) == 1: alone[name(rows[next(iter(live))])] += t - last
    last = t
    if kind: live.add(i)
    else: live.discard(i)
wall = max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[name(r)]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("per step (%g steps): wall %.3f ms, some kernel running %.3f ms (%.1f %%), idle %.3f ms; summed durations %.3f ms" %
      (steps, wall / 1e6 / steps, union / 1e6 / steps, 100.0 * union / wall, (wall - union) / 1e6 / steps, sum(a[1] for a in agg.values()) / 1e6 / steps))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-28s n/step %7.1f  sum %8.3f ms/step  avg %8.2f us  alone %8.3f ms/step" % (k[:28], a[0] / steps, a[1] / 1e6 / steps, a[1] / a[0] / 1e3, alone[k] / 1e6 / steps))
short = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "k_sweep_shell" in r["Kernel_Name"]]
for lim in (10e3, 20e3, 50e3, 100e3, 1e9):
    sel = [d for d in short if d < lim]
    print("shell launches shorter than %6.0f us: %6.1f per step, %7.3f ms per step" % (lim / 1e3, len(sel) / steps, sum(sel) / 1e6 / steps))
