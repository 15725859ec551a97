mkdir -p gpurun_out/r05zz
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05zz/smoke.log 2>&1; tail -1 gpurun_out/r05zz/smoke.log
python -m pytest tests -x -q -m gpu > gpurun_out/r05zz/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05zz/pytest.log; tail -4 gpurun_out/r05zz/pytest.log
python bench.py > gpurun_out/r05zz/bench_default.json 2> gpurun_out/r05zz/bench_default.err; tail -c 400 gpurun_out/r05zz/bench_default.json
