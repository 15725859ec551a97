#!/bin/bash
# Development probe of the plane-binned kernel: kernel trace (per-launch durations in launch order) and the SQ counters of one
# bench pass.   profiles/r04_plane/probe.sh <tag> [env assignments...]   ->  gpurun_out/<tag>/
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/probe_$TAG; rm -rf "$W"; mkdir -p "$W"
FL="--steps 1 --warmup 0 --no-cpu-baseline --no-other-mode --no-small-leg"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$W/kt" -o kt -- python3 bench.py $FL > "$OUT/kt_bench.json" 2> "$OUT/kt.err"
python3 - "$(find "$W/kt" -name '*kernel_trace.csv' | head -1)" > "$OUT/trace_summary.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
seq = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    if "k_sweep_" in n: seq.append((n, d))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-110s n=%5d total=%9.3f ms avg=%8.4f ms" % (n[:110], c, t, t / c))
# the last pass: launches in order
tail = seq[-260:]
print("--- last launches in order (kernel, ms)")
for n, d in tail:
    print("%s %.4f" % ("P" if ("plane" in n or "tile" in n) else ("F" if "fused" in n else "S"), d))
PY
pmc () { local tag=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$W/$tag" -o p -- python3 bench.py $FL > /dev/null 2> "$OUT/pmc_$tag.err"
  python3 - "$(find "$W/$tag" -name '*counter_collection.csv' | head -1)" >> "$OUT/pmc_summary.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if "k_sweep_" not in n: continue
    n = n.split("<")[0] if False else n
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, c in agg.items():
    print(n[:100], {k: "%.4g" % v for k, v in c.items()})
PY
}
: > "$OUT/pmc_summary.txt"
pmc SQ SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY
pmc SQ2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES
pmc FETCH FETCH_SIZE
pmc WRITE WRITE_SIZE
pmc GRBM GRBM_COUNT GRBM_GUI_ACTIVE
cat "$OUT/trace_summary.txt" | head -14; cat "$OUT/pmc_summary.txt"
