mkdir -p gpurun_out/r05m
python -m pytest tests/test_gpu_xray.py tests/test_gpu_thermal.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05m/pytest.log; tail -4 gpurun_out/r05m/pytest.log
for mode in exact fast; do python tests/fuzz_gpu.py 200 60000 $mode > gpurun_out/r05m/fuzz_$mode.txt 2>&1; tail -1 gpurun_out/r05m/fuzz_$mode.txt; done
python bench.py --deterministic --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-dropin-leg --no-mix-ceiling 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('deterministic: ms_per_step', d['ms_per_step'], 'launch ms', d['roofline']['avg_launch_ms'])"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-dropin-leg --no-mix-ceiling 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('atomics: ms_per_step', d['ms_per_step'])"
