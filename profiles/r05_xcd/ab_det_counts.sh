#!/bin/bash
# ordered rates (deterministic_rates = 1) with and without the plane-ordered mapping at several source counts, 256^3
run () { env $1 python bench.py --deterministic --sources $2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-18s sources %4d  ms_per_step %8.2f  sum_nbox %d  phih_sum %.12e' % ('$1', $2, d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum']))"; }
for S in 64 125 250 1000; do for rep in 1 2; do run C2R_XCD_ORDER=0 $S; run C2R_XCD_ORDER=1 $S; done; done
