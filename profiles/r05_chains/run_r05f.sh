mkdir -p gpurun_out/r05f
python -m pytest tests/test_gpu_native_ranks.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_two_ranks.py tests/test_gpu_few_sources.py tests/test_gpu_api.py tests/test_gpu_chains.py -x -q -m gpu > gpurun_out/r05f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05f/pytest.log; tail -6 gpurun_out/r05f/pytest.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C2R_PROFILE_TCC=1 timeout 1500 bash profiles/run_profile.sh r05_final > gpurun_out/r05f/profile.log 2>&1; tail -8 gpurun_out/r05f/profile.log
