cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k replayed 2>&1 | tail -5
timeout 1500 python tests/fuzz_chains_gpu.py 0 200 1 > gpurun_out/fuzz_chains_fast.txt 2>&1; tail -3 gpurun_out/fuzz_chains_fast.txt
timeout 1500 python tests/fuzz_chains_gpu.py 0 200 0 > gpurun_out/fuzz_chains_exact.txt 2>&1; tail -3 gpurun_out/fuzz_chains_exact.txt
