#!/bin/bash
# polling the sub-box count events before blocking on them (C2R_POLL_WAIT=1, shipped) against hipEventSynchronize alone (=0)
for rep in 1 2 3; do
  for S in 48 125 1000; do
    for v in 1 0; do
      steps=10; [ $S = 1000 ] && steps=3
      C2R_POLL_WAIT=$v python bench.py --sources $S --steps $steps --warmup 2 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('poll=$v S=$S ms_per_step %.3f' % d['ms_per_step'])"
    done
  done
done
