mkdir -p gpurun_out/r05end
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05end/smoke.log 2>&1; tail -1 gpurun_out/r05end/smoke.log
python -m pytest tests -x -q -m gpu > gpurun_out/r05end/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05end/pytest.log; tail -4 gpurun_out/r05end/pytest.log
python bench.py > gpurun_out/r05end/bench_default.json 2> gpurun_out/r05end/bench_default.err; tail -c 400 gpurun_out/r05end/bench_default.json
