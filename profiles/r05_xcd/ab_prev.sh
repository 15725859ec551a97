#!/bin/bash
# the previous commit's library (ab_old/pkg, built from `git show HEAD:...`) against the working tree's, alternating on one box
run () { env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-30s ms_per_step %8.2f  sum_nbox %d  phih_sum %.12e  launch_ms %.4f' % ('$1'[-28:], d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum'], d['roofline']['avg_launch_ms']))"; }
for rep in 1 2 3 4; do
  run C2RAY_HIP_LIB=$PWD/ab_old/pkg/libc2ray_hip.so "$*"
  run C2R_NOP=1 "$*"
done
