mkdir -p gpurun_out/r05d
python -m pytest tests -x -q -m gpu > gpurun_out/r05d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05d/pytest.log; tail -4 gpurun_out/r05d/pytest.log
python bench.py > gpurun_out/r05d/bench_default.json 2> gpurun_out/r05d/bench_default.err; tail -c 1500 gpurun_out/r05d/bench_default.json
for mode in 1 0; do
  C2R_SWEEP_MODE=$mode python profiles/steps_schedule.py --mesh 256 --sources 1000 --steps 14 > gpurun_out/r05d/schedule_1000_mode$mode.jsonl 2>&1
  C2R_SWEEP_MODE=$mode python profiles/steps_schedule.py --mesh 256 --sources 100 --steps 14 > gpurun_out/r05d/schedule_100_mode$mode.jsonl 2>&1
done
tail -1 gpurun_out/r05d/schedule_*.jsonl
