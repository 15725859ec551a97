#!/bin/bash
# ab_variant.sh for a source count: ab_variant_s.sh <tag> <sources> <steps>
tag=$1; S=$2; K=$3
run () { env $1 python bench.py --sources $S --steps $K --warmup 2 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-40s sources %4d ms_per_step %8.3f  sum_nbox %d' % ('$1'[-36:], $S, d['ms_per_step'], d['check']['sum_nbox_last_step']))"; }
for rep in 1 2 3; do
  run C2RAY_HIP_LIB=$PWD/c2-ray3dm_amd/libc2ray_hip_$tag.so
  run C2R_NOP=1
done
