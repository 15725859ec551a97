mkdir -p gpurun_out/r05c
python -m pytest tests -x -q -m gpu > gpurun_out/r05c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05c/pytest.log; tail -4 gpurun_out/r05c/pytest.log
profiles/micro/ab_chains.sh gpurun_out/r05c/ab_chains.txt > gpurun_out/r05c/ab.log 2>&1; cat gpurun_out/r05c/ab_chains.txt
python bench.py > gpurun_out/r05c/bench_default.json 2> gpurun_out/r05c/bench_default.err; tail -c 3000 gpurun_out/r05c/bench_default.json
