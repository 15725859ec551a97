cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --mesh 504 --sources 10000 --density lognormal --no-cpu-baseline --no-other-mode --no-mix-ceiling --steps 2 --warmup 1 > gpurun_out/bench_504_10000.json 2> gpurun_out/bench_504_10000.err
python bench.py --mesh 504 --sources 1250 --density lognormal --no-cpu-baseline --no-other-mode --no-mix-ceiling --steps 2 --warmup 1 > gpurun_out/bench_504_1250.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/bench_504_10000.json","gpurun_out/bench_504_1250.json"):
    j=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], j["config"]["mean_subboxes_per_source"][-1])
PY
tail -2 gpurun_out/bench_504_10000.err
