cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 1 0; do export C2R_TILED=$t
for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pq; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pq -o p -- python3 profiles/micro/ablate.py 1 > /tmp/abl.txt 2>&1
python3 - $c $t <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void c2r::","")
    if k.startswith("k_sweep"): tot[k] += float(r["Counter_Value"])
vis = 2 * 1.6777e10      # warm-up pass + 1 timed pass
print("tiled=%s %s KiB by kernel:" % (sys.argv[2], sys.argv[1]), {k: "%.3g" % v for k, v in tot.items()}, "bytes/visit (raw) %.2f" % (sum(tot.values()) * 1024 / vis))
PY
done; done
