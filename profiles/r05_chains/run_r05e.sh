mkdir -p gpurun_out/r05e
python -m pytest tests -x -q -m gpu > gpurun_out/r05e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05e/pytest.log; tail -4 gpurun_out/r05e/pytest.log
python bench.py > gpurun_out/r05e/bench_default.json 2> gpurun_out/r05e/bench_default.err; tail -c 1200 gpurun_out/r05e/bench_default.json
