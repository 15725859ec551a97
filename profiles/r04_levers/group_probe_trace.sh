#!/bin/bash
export TMPDIR=/tmp
W=/tmp/gp2; rm -rf $W; mkdir -p $W
sed -i 's/for S in (1000, 250, 96, 48, 32, 16):/for S in (1000, 48):/' profiles/r04_levers/group_probe.py
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $W -o kt -- python3 profiles/r04_levers/group_probe.py > /dev/null 2> $W/err
python3 - "$(find $W -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows if "k_sweep_shell_fast" in r["Kernel_Name"]]
print(len(d))
qs = list(range(11, 129))
for name, S, seg in (("S=1000", 1000, d[-354:-236]), ("S=48", 48, d[-118:])):
    for q0, q1 in ((11, 32), (32, 48), (48, 64), (64, 80), (80, 100), (100, 120), (120, 129)):
        t = sum(x for q, x in zip(qs, seg) if q0 <= q < q1); cells = sum(24 * q * q + 2 for q in qs if q0 <= q < q1)
        print("%s  q %3d..%3d  %8.3f ms  %.4f ns per (cell, source)" % (name, q0, q1 - 1, t, 1e6 * t / (cells * S)))
PY
