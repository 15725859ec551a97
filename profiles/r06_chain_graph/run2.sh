cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_t2.log
python bench.py > gpurun_out/r6_bench2.json 2> gpurun_out/r6_bench2.err
tail -3 gpurun_out/r6_bench2.err
