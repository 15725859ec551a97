cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_chains.py tests/test_gpu_api.py tests/test_gpu_few_sources.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_t1.log
for opt in "chain_graph=0" "chain_graph=1" "chain_graph=1 --option chains=3" "chain_graph=1 --option chains=4" "chain_graph=0" "chain_graph=1"; do
  echo "== $opt" >> gpurun_out/r6_share.log
  python bench.py --sources 125 --steps 10 --warmup 4 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --option $opt 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], j['config']['mean_subboxes_per_source'][-1])
    else: print(l.rstrip()[-300:])
" >> gpurun_out/r6_share.log
done
