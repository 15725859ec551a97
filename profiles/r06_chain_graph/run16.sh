cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LOG=gpurun_out/r6_ab_lds.log; : > $LOG
one () { python bench.py "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'])
"; }
NOLEG="--no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
for rep in 1 2 3; do
  for lib in "" "$PWD/c2-ray3dm_amd/libc2ray_hip_nolds.so"; do
    export C2RAY_HIP_LIB=$lib; [ -z "$lib" ] && unset C2RAY_HIP_LIB
    echo "== lib=${lib:-default(LDS planes)}" >> $LOG
    echo -n "128^3 x 1: " >> $LOG;  one --mesh 128 --sources 1 --steps 400 --warmup 20 $NOLEG >> $LOG
    echo -n "128^3 x 4: " >> $LOG;  one --mesh 128 --sources 4 --steps 300 --warmup 20 $NOLEG >> $LOG
    echo -n "cold 256^3 x 1000: " >> $LOG; one --x-init 2e-4 --steps 40 --warmup 5 $NOLEG >> $LOG
    echo -n "cold 256^3 x 125: " >> $LOG; one --x-init 2e-4 --sources 125 --steps 60 --warmup 5 $NOLEG >> $LOG
    echo -n "headline: " >> $LOG; one --steps 5 --warmup 2 $NOLEG >> $LOG
  done
done
cat $LOG
