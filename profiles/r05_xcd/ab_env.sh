#!/bin/bash
# same-box A/B of one environment switch of the library: ab_env.sh VAR=a VAR=b [bench flags]
A=$1; B=$2; shift 2
run () { env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-28s %-14s ms_per_step %8.2f  sum_nbox %d  phih_sum %.12e  launch_ms %.4f' % ('$1', '$2', d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum'], d['roofline']['avg_launch_ms']))"; }
for rep in 1 2 3; do
  run $A "$*"
  run $B "$*"
done
