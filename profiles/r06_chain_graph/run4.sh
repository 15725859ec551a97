cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL="--sources 125 --steps 10 --warmup 5 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
run () { echo "== $*" >> gpurun_out/r6_share4.log; python bench.py $FL "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], j['config']['mean_subboxes_per_source'][-1])
" >> gpurun_out/r6_share4.log; }
for rep in 1 2; do
run --option chain_graph=1
run --option chain_graph=1 --option xcd_order=1 --option xcd_min_sources=16
run --option chain_graph=1 --option xcd_order=1 --option xcd_min_sources=16 --option chains=1
run --option chain_graph=1 --option chains=1
run --option chain_graph=1 --option stream_hint=0
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('headline', j['ms_per_step'], j['ms_per_step']/8)
" >> gpurun_out/r6_share4.log
cat gpurun_out/r6_share4.log
