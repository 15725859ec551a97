mkdir -p gpurun_out/r05h
python -m pytest tests -x -q -m gpu > gpurun_out/r05h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05h/pytest.log; tail -4 gpurun_out/r05h/pytest.log
python bench.py > gpurun_out/r05h/bench_default.json 2> gpurun_out/r05h/bench_default.err; tail -c 600 gpurun_out/r05h/bench_default.json
python bench.py --thermal --no-small-leg --no-dropin-leg > gpurun_out/r05h/bench_thermal.json 2> gpurun_out/r05h/bench_thermal.err
for mode in exact fast; do
  python tests/fuzz_gpu.py 300 40000 $mode > gpurun_out/r05h/fuzz_$mode.txt 2>&1; tail -1 gpurun_out/r05h/fuzz_$mode.txt
  python tests/_fuzz_steps.py 100 50000 $mode > gpurun_out/r05h/step_fuzz_$mode.txt 2>&1; tail -1 gpurun_out/r05h/step_fuzz_$mode.txt
done
