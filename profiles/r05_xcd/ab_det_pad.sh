#!/bin/bash
# ordered rates: per-source grids at a stride of exactly N^3 f64 (C2R_GBOX_PAD=0: 2^27 B at 256^3) against a padded stride
# (default 520 elements), with the plain grid and with the plane-ordered mapping
run () { env $1 python bench.py --deterministic --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-36s ms_per_step %8.2f  sum_nbox %d  phih_sum %.12e' % ('$1', d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum']))"; }
for rep in 1 2; do
  run "C2R_GBOX_PAD=0 C2R_XCD_ORDER=0"; run "C2R_GBOX_PAD=520 C2R_XCD_ORDER=0"; run "C2R_GBOX_PAD=0"; run "C2R_GBOX_PAD=520"
done
