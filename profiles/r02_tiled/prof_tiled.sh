cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export C2R_TILED=1
rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python3 profiles/micro/ablate.py 2 > /dev/null 2>&1
head -8 $(find /tmp/pt -name "*kernel_stats.csv" | head -1) | cut -c1-150
rm -rf /tmp/pq; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d /tmp/pq -o p -- python3 profiles/micro/ablate.py 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void c2r::","")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    if c.get("SQ_WAVES", 0) > 1e5:
        w = c["SQ_WAVES"]
        print(k, "waves %.3g" % w, "VALU/wave %.0f" % (c["SQ_INSTS_VALU"]/w), "SALU/wave %.0f" % (c["SQ_INSTS_SALU"]/w), "LDS/wave %.0f" % (c["SQ_INSTS_LDS"]/w),
              "wave quad-cycles %.0f" % (c["SQ_WAVE_CYCLES"]/w), "wait_any %.2f" % (c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"]), "wait_inst %.2f" % (c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]), "valu active quad %.0f" % (c["SQ_ACTIVE_INST_VALU"]/w))
PY
