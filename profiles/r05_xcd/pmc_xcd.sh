#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of one bench pass with the plain grid (C2R_XCD_ORDER=0) and the plane-ordered
# mapping (=1): sums over the sweep kernels' dispatches.  Separate --pmc passes, no trace domains.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_xcd; mkdir -p "$OUT"
for mode in 0 1; do
  export C2R_XCD_ORDER=$mode
  for tag in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    name=$(echo $tag | cut -d' ' -f1)
    W=/tmp/pmcx_${mode}_$name; rm -rf "$W"
    timeout 600 rocprofv3 --pmc $tag --output-format csv -d "$W" -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-mode --no-small-leg --no-dropin-leg --no-mix-ceiling > "$OUT/bench_$mode.json" 2> "$OUT/err_${mode}_$name.txt"
    python3 - "$(find "$W" -name '*counter_collection.csv' | head -1)" $mode <<'PY' | tee -a "$OUT/summary.txt"
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("<")[0]
    if "k_sweep_shell_fast" not in k: continue
    tot[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(tot.items()):
    print("XCD_ORDER=%s %-40s %-24s sum %.6e over %d dispatches" % (sys.argv[2], k, c, v, n[(k, c)]))
PY
  done
done
