#!/bin/bash
# Same-box A/B of the XCD-aware, plane-ordered block mapping of the far shells (k_sweep_shell_fast_xcd) against the plain
# (tile, face, source) grid, 256^3 x 1000 sources; C2R_XCD_ORDER = 0 never / 1 always where it can run.
run () { env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-18s %-22s ms_per_step %8.2f  sum_nbox %d  phih_sum %.12e  launch_ms %.4f' % ('$1', '$2', d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum'], d['roofline']['avg_launch_ms']))"; }
for rep in 1 2 3; do
  run C2R_XCD_ORDER=0 "$*"
  run C2R_XCD_ORDER=1 "$*"
done
