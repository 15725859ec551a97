#!/bin/bash
# 384 - 768 sources at 256^3: two chains in flight (the default there, no plane-ordered mapping: a chain holds half the sources)
# against one chain with the plane-ordered mapping
run () { env $1 python bench.py --sources $2 --steps 4 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-34s sources %4d  ms_per_step %8.2f  sum_nbox %d' % ('$1', $2, d['ms_per_step'], d['check']['sum_nbox_last_step']))"; }
for rep in 1 2; do
  for S in 400 500 640 768; do
    run C2R_NOP=1 $S
    run C2R_CHAINS=1 $S
    run "C2R_CHAINS=1 C2R_XCD_ORDER=0" $S
  done
done
