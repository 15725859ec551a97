cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_chains.py -x -q -m gpu -k "non_isothermal" 2>&1 | tail -15
