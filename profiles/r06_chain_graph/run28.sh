cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_chains.py -x -q -m gpu -k "on_request" 2>&1 | tail -15
