#!/usr/bin/env python3
"""Timeline of one bench.py run from a rocprofv3 kernel trace: per kernel name the launches, busy time and the idle gap
in front of each launch; totals of GPU-busy vs wall over the timed region (the last `frac` of the trace).
    python profiles/timeline.py <kernel_trace.csv> [frac=0.5]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for r in rows:
    k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg[k]; a[0] += 1; a[1] += e - s
    if prev_end is not None: a[2] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(a[1] for a in agg.values())
print("launches %d  wall %.3f ms  busy %.3f ms (%.0f %%)  mean gap %.2f us" % (len(rows), wall / 1e6, busy / 1e6, 100 * busy / wall, (wall - busy) / 1e3 / len(rows)))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print("%-34s n %6d  avg %7.2f us  gap before %7.2f us" % (k[-34:], a[0], a[1] / a[0] / 1e3, a[2] / a[0] / 1e3))
