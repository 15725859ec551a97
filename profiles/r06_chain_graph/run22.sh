cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_native_ranks.py -x -q -m gpu 2>&1 | tail -12
