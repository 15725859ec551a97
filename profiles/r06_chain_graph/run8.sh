cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_final
python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_final/pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final/smoke.log 2>&1
bash profiles/run_profile.sh r06_final > gpurun_out/r06_final_run.log 2>&1
python bench.py > gpurun_out/r06_final/bench_default.json 2> gpurun_out/r06_final/bench_default.err
cat gpurun_out/r06_final/pytest_gpu_full.log gpurun_out/r06_final/smoke.log; tail -8 gpurun_out/r06_final_run.log | cut -c1-600
