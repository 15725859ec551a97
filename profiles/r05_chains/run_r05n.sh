mkdir -p gpurun_out/r05n
python -m pytest tests/test_gpu_parity.py tests/test_gpu_thermal.py tests/test_gpu_few_sources.py tests/test_gpu_native_ranks.py tests/test_gpu_two_ranks.py tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r05n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05n/pytest.log; tail -4 gpurun_out/r05n/pytest.log
for i in 1 2; do
python bench.py --deterministic --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-dropin-leg --no-mix-ceiling 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('deterministic: ms_per_step', d['ms_per_step'], 'launch ms', d['roofline']['avg_launch_ms'])"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-dropin-leg --no-mix-ceiling 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('atomics: ms_per_step', d['ms_per_step'])"
done
