cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_allfrac.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_dropin_pieces.py tests/test_gpu_percell.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r6_t7.log
cat gpurun_out/r6_t7.log
