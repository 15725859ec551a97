cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL="--sources 125 --steps 10 --warmup 5 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
LOG=gpurun_out/r6_ab11.log; : > $LOG
run () { echo "== $*" >> $LOG; python bench.py $FL "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'])
" >> $LOG; }
for rep in 1 2; do
run --option chains=2
run --option chains=3
run --option chains=4 --option pair_shells=0
run --option chains=4
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('headline', j['ms_per_step'], j['ms_per_step']/8)
" >> $LOG
cat $LOG
