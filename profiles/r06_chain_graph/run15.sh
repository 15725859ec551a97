cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_few_sources.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_thermal.py tests/test_gpu_xray.py tests/test_gpu_chains.py tests/test_gpu_xcd_order.py tests/test_gpu_allfrac.py -x -q -m gpu 2>&1 | tail -8
