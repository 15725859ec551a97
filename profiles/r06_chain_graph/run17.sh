cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL="--sources 125 --steps 10 --warmup 5 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
LOG=gpurun_out/r6_ab_stagger.log; : > $LOG
run () { echo -n "$* : " >> $LOG; python bench.py $FL "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'])
" >> $LOG; }
for rep in 1 2 3; do
for us in 0 10 25 50 100 200; do run --option chain_stagger_us=$us; done
run --option chain_stagger_us=50 --option chains=3
done
cat $LOG
