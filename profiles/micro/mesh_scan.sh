# roofline fraction of the sweep kernel against the mesh size at a fixed number of sources: mesh_scan.sh [sources]
for n in 128 256 384 504; do
  python bench.py --mesh $n --sources ${1:-300} --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode 2>/dev/null | tail -1 > /tmp/ms.json
  python - <<PY
import json
j = json.load(open("/tmp/ms.json")); r = j["roofline"]
print("$n^3: %.2f ms/step  visited/s %.3e  frac %.3f  sub-boxes/source %.1f" % (j["ms_per_step"], j["config"]["visited_per_s"], r["frac"] or 0, j["config"]["mean_subboxes_per_source"][-1]))
PY
done
