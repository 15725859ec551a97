cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL="--sources 125 --steps 10 --warmup 5 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
LOG=gpurun_out/r6_ab10.log; : > $LOG
run () { echo "== $*" >> $LOG; python bench.py $FL "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'])
" >> $LOG; }
for rep in 1 2 3; do
run --option chain_graph=0
run --option chain_graph=1 --option chain_tail=0
run --option chain_graph=1 --option chain_tail=1
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('headline', j['ms_per_step'], j['ms_per_step']/8)
" >> $LOG
rocm-smi --showclocks 2>/dev/null | head -20 >> $LOG
cat $LOG
