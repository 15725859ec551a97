cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_final
python -X faulthandler -m pytest tests -v -m gpu 2>&1 > gpurun_out/r06_final/pytest_gpu_verbose.log
grep -n "Fatal\|Segmentation\|Aborted\|PASSED\|FAILED\|ERROR" gpurun_out/r06_final/pytest_gpu_verbose.log | tail -5
grep -n "Fatal Python error" -A30 gpurun_out/r06_final/pytest_gpu_verbose.log | head -60
tail -5 gpurun_out/r06_final/pytest_gpu_verbose.log | cut -c1-300
