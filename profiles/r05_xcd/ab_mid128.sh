#!/bin/bash
# 128^3: chains in flight (default for 64 - 768 sources) against one chain with the plane-ordered mapping (from 192 sources there)
run () { env $1 python bench.py --mesh 128 --sources $2 --steps 6 --warmup 2 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-34s mesh 128 sources %4d  ms_per_step %8.3f  sum_nbox %d' % ('$1', $2, d['ms_per_step'], d['check']['sum_nbox_last_step']))"; }
for rep in 1 2; do
  for S in 200 300 500 768; do
    run C2R_NOP=1 $S
    run C2R_CHAINS=1 $S
    run "C2R_CHAINS=1 C2R_XCD_ORDER=0" $S
  done
done
