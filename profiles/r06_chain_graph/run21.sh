cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err; echo "rc=$?"
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/bench_head.json") if l.startswith("{")][-1])
print(j["ms_per_step"], j["roofline"]["frac"], j["one_gpu_share_of_8"]["ratio"], j["share_504"], j["other_sweep_mode"]["roofline"]["frac"], j["parity"]["integers_equal"], list(j.keys()))
PY
python -m pytest tests/test_gpu_bench_two_ranks.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -3
