cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_chains.py tests/test_gpu_few_sources.py tests/test_gpu_thermal.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_t6.log
cat gpurun_out/r6_t6.log
FL="--sources 125 --steps 10 --warmup 5 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
LOG=gpurun_out/r6_hostload2.log; : > $LOG
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
