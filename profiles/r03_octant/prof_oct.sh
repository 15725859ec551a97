#!/bin/bash
# kernel trace + SQ counters of frozen-state passes (profiles/micro/ablate.py) for one C2R_OCTANT level:
#   profiles/r03_strip/prof_strip.sh <name> <strip level> [extra env assignments...]
set -u
NAME=$1; LEVEL=$2; shift 2
for kv in "$@"; do export "$kv"; done
export C2R_OCTANT=$LEVEL TMPDIR=/tmp
OUT=$PWD/gpurun_out/$NAME; mkdir -p "$OUT"
W=/tmp/prof_$NAME; rm -rf "$W"; mkdir -p "$W"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$W/kt" -o kt -- python3 profiles/micro/ablate.py 2 > "$OUT/kt.txt" 2> "$OUT/kt.err"
cp "$(find "$W/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
pmc () { local tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$W/$tag" -o p -- python3 profiles/micro/ablate.py 1 > "$OUT/pmc_$tag.txt" 2> "$OUT/pmc_$tag.err"
  python3 - "$(find "$W/$tag" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$tag.json" <<'PY'
import csv, sys, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[k] += 1
json.dump({k: dict(v, dispatches=n[k]) for k, v in acc.items() if "sweep" in k}, open(sys.argv[2], "w"), indent=1)
PY
}
pmc SQ SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY
pmc SQ2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_LEVEL_WAVES
if [ "${PROF_TRAFFIC:-0}" = 1 ]; then pmc FETCH_SIZE FETCH_SIZE; pmc WRITE_SIZE WRITE_SIZE; fi
head -12 "$OUT/kernel_stats.csv" | cut -d, -f1-8 | cut -c1-200
cat "$OUT"/pmc_SQ.json "$OUT"/pmc_SQ2.json
