#!/usr/bin/env python3
"""Per-shell efficiency of ONE chain from a rocprofv3 kernel trace of `bench.py --sources S --option chains=1 --option chain_graph=0`:
the shell launches of the last step in order (q = 11 ...), each launch's duration, the gap in front of it, and the time the same
visits take at the rate of the 1000-source step (ps per visit given on the command line).
    python profiles/per_shell_efficiency.py <kernel_trace.csv> <sources> <mesh> <ps_per_visit_at_1000_sources>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
S, n, ps = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
sh = [r for r in rows if "k_sweep_shell" in r["Kernel_Name"]]
per_step = (n // 2 - 10)                      # shells 11 .. n/2 (the first ten run in the fused kernel)
last = sh[-per_step:]
allk = rows[rows.index(last[0]):]
tot_d = tot_i = tot_g = 0.0
print("   q  blocks  rounds  dur_us  gap_us  ideal_us  dur/ideal")
prev_end = None
for k, r in enumerate(last):
    q = 11 + k
    side = min(2 * q + 1, n)
    cells = 6 * side * side if 2 * q + 1 <= n else 0
    # cells of shell q inside the mesh: the cube surface clipped to -n/2 .. n/2-1 (even n): use the surface for q < n/2
    cells = 24 * q * q + 2 if q < n // 2 else (n ** 3 - (n - 1) ** 3)
    rows_per = (side + 2) // 3
    tiles = (side * rows_per + 255) // 256
    blocks = S * 6 * tiles
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    i = cells * S * ps * 1e-6
    g = 0.0 if prev_end is None else (int(r["Start_Timestamp"]) - prev_end) / 1e3
    prev_end = int(r["End_Timestamp"])
    tot_d += d; tot_i += i; tot_g += g
    if q % 8 == 3 or q < 14:
        print("%4d %7d %7.2f %7.1f %7.1f %9.1f %9.2f" % (q, blocks, blocks / 1792.0, d, g, i, d / i))
print("sum of %d shell launches: %.3f ms, ideal %.3f ms, gaps between them (incl. the small kernels) %.3f ms" % (len(last), tot_d / 1e3, tot_i / 1e3, tot_g / 1e3))
