cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LOG=gpurun_out/r6_ab_fusedfew.log; : > $LOG
one () { python bench.py "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], j['check']['sum_nbox_last_step'])
"; }
NOLEG="--no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
for rep in 1 2 3; do
  for v in 1 2 0; do
    echo "== fused_boxes_few=$v" >> $LOG
    echo -n "128^3 x 1: " >> $LOG;  one --mesh 128 --sources 1 --steps 400 --warmup 20 $NOLEG --option fused_boxes_few=$v >> $LOG
    echo -n "128^3 x 4: " >> $LOG;  one --mesh 128 --sources 4 --steps 300 --warmup 20 $NOLEG --option fused_boxes_few=$v >> $LOG
    echo -n "256^3 x 16: " >> $LOG;  one --mesh 256 --sources 16 --steps 100 --warmup 10 $NOLEG --option fused_boxes_few=$v >> $LOG
  done
done
cat $LOG
