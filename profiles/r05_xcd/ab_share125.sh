#!/bin/bash
# one GPU's share of the 8-GPU bench (125 sources, chains) with the library of the commit before the plane-ordered mapping
# (ab_old/pkg/libc2ray_hip.so, built from f7f1a85) against HEAD's, alternating on one box
run () { env $1 python bench.py --sources 125 --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-50s ms_per_step %8.3f  sum_nbox %d' % ('$1', d['ms_per_step'], d['check']['sum_nbox_last_step']))"; }
for rep in 1 2 3; do
  run C2RAY_HIP_LIB=$PWD/ab_old/pkg/libc2ray_hip.so
  run C2R_XCD_ORDER=-1
done
