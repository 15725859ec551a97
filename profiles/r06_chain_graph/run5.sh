# The 125-source share under a starved host: the bench process (and every runtime thread it starts) pinned to ONE core that two busy
# loops also run on -- chains driven launch by launch against chains replayed as one captured sequence each.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL="--sources 125 --steps 10 --warmup 5 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
LOG=gpurun_out/r6_hostload.log; : > $LOG
run () { echo "== $*" >> $LOG; "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'])
" >> $LOG; }
for rep in 1 2; do
  echo "-- quiet host" >> $LOG
  run python bench.py $FL --option chain_graph=0
  run python bench.py $FL --option chain_graph=1
  echo "-- bench pinned to core 5 together with two busy loops" >> $LOG
  taskset -c 5 python -c "while True: pass" & P1=$!
  taskset -c 5 python -c "while True: pass" & P2=$!
  run taskset -c 5 python bench.py $FL --option chain_graph=0
  run taskset -c 5 python bench.py $FL --option chain_graph=1
  kill $P1 $P2; wait $P1 $P2 2>/dev/null
done
cat $LOG
