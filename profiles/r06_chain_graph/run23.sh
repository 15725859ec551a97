cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_final
python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_final/pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final/smoke.log 2>&1
cat gpurun_out/r06_final/pytest_gpu_full.log gpurun_out/r06_final/smoke.log
