#!/bin/bash
# per-launch durations of the far shells for S sources (kernel trace), to compare per-source cost with S = 1000
export TMPDIR=/tmp
for S in 24 48 1000; do
W=/tmp/mp_$S; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $W -o kt -- python3 bench.py --sources $S --steps 1 --warmup 0 --no-cpu-baseline --no-other-mode --no-small-leg > /dev/null 2> $W/err
python3 - "$(find $W -name '*kernel_trace.csv' | head -1)" $S <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); S = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows if "k_sweep_shell_fast" in r["Kernel_Name"]]
last = d[-118:]          # the last pass: shells 11..128
qs = list(range(11, 129))
for q0, q1 in ((48, 64), (64, 80), (80, 100), (100, 120), (120, 129)):
    t = sum(x for q, x in zip(qs, last) if q0 <= q < q1); cells = sum(24 * q * q + 2 for q in qs if q0 <= q < q1)
    print("S=%4d  q %3d..%3d  %.3f ms  %.3f ns per (cell, source)" % (S, q0, q1 - 1, t, 1e6 * t / (cells * S)))
PY
done
