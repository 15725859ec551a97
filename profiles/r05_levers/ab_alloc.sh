#!/bin/bash
# allocation flags of the streaming arrays: shell planes / n_HI as plain hipMalloc (default), uncached (MTYPE UC) or fine-grained
run () { env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$1 ms_per_step %.2f  sum_nbox %d  phih_sum %.12e' % (d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum']))"; }
for rep in 1 2; do
  run C2R_PLANES_ALLOC=0
  run C2R_PLANES_ALLOC=1
  run C2R_PLANES_ALLOC=2
  run "C2R_PLANES_ALLOC=1 C2R_NHI_ALLOC=1"
  run C2R_NHI_ALLOC=1
done
