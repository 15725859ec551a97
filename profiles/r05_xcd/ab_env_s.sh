#!/bin/bash
# same-box A/B of one environment switch at a source count: ab_env_s.sh VAR=a VAR=b <sources> <steps>
A=$1; B=$2; S=$3; K=$4
run () { env $1 python bench.py --sources $S --steps $K --warmup 2 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-26s sources %4d ms_per_step %8.3f  sum_nbox %d  phih_sum %.12e' % ('$1', $S, d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum']))"; }
for rep in 1 2 3 4; do
  run $A
  run $B
done
