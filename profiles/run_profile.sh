#!/bin/bash
# One profile set of bench.py on the GPU box (the recipe of profiles/README.md):
#   profiles/run_profile.sh <name> [bench.py flags...]      ->  gpurun_out/<name>/{bench.json,kernel_stats.csv,kernel_trace.csv,
#                                                               kt_bench.json,pmc_*.csv,pmc_bench.json,traffic.json}
# Counters are collected in separate --pmc passes with nothing but the counter list (no trace domains).
set -u
NAME=$1; shift
OUT=$PWD/gpurun_out/$NAME
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/prof_$NAME; rm -rf "$W"; mkdir -p "$W"
python3 bench.py --steps 3 --warmup 1 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$W/kt" -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-configs-leg --no-dropin-leg --no-mix-ceiling "$@" > "$OUT/kt_bench.json" 2> "$OUT/kt.err"
cp "$(find "$W/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
cp "$(find "$W/kt" -name '*kernel_trace.csv' | head -1)" "$OUT/kernel_trace.csv"
pmc () {   # $1 = tag, rest = counters
  local tag=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$W/$tag" -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-mode --no-small-leg --no-configs-leg --no-dropin-leg --no-mix-ceiling $BENCH_FLAGS > "$OUT/pmc_bench.json" 2> "$OUT/pmc_$tag.err"
  cp "$(find "$W/$tag" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$tag.csv"
}
BENCH_FLAGS="$*"
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY
pmc SQ2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_LEVEL_WAVES
pmc GRBM GRBM_COUNT GRBM_GUI_ACTIVE
if [ "${C2R_PROFILE_TCC:-1}" = 1 ]; then      # request-size split behind FETCH_SIZE / WRITE_SIZE, L2 hit rate
  pmc TCC1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  pmc TCC2 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum
  pmc TCC3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  pmc TCC4 TCC_WRITE_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum
  pmc TCP1 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
fi
# keep the trace small: only what summarize.py needs stays (the per-launch rows of the sweep kernel)
python3 - "$OUT" <<'PY'
import csv, sys, os
d = sys.argv[1]
p = os.path.join(d, "kernel_trace.csv")
rows = list(csv.DictReader(open(p)))
keep = [r for r in rows if "k_sweep_shell" in r["Kernel_Name"]]
with open(p, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
    w.writeheader()
    for r in keep:
        w.writerow({"Kernel_Name": r["Kernel_Name"].split("(")[0], "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"]})
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2", "GRBM", "TCC1", "TCC2", "TCC3", "TCC4", "TCP1"):
    p = os.path.join(d, "pmc_%s.csv" % tag)
    if not os.path.exists(p): continue
    rows = list(csv.DictReader(open(p)))
    cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(p, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        for r in rows:
            r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
            w.writerow({c: r[c] for c in cols})
PY
python3 profiles/summarize.py "$OUT" > /dev/null
python3 -c "import json,sys; t=json.load(open(sys.argv[1])); [print(k, json.dumps(t.get(k))) for k in ('sweep_kernel','sweep_bytes_per_visit','agreement','sweep_sq','sweep_sq2_per_wave','sweep_counters_per_visit')]" "$OUT/traffic.json"
