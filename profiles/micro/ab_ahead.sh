#!/bin/bash
# how far ahead of the device the lock-step chain schedule runs (C2R_CHAIN_AHEAD: variant builds libc2ray_hip_ahead{1,3}.so
# against the shipped 2), 125 and 250 sources, alternating on one box
for rep in 1 2; do
  for lib in "" ahead1 ahead3; do
    for S in 125 250; do
      if [ -z "$lib" ]; then unset C2RAY_HIP_LIB; else export C2RAY_HIP_LIB=$PWD/c2-ray3dm_amd/libc2ray_hip_$lib.so; fi
      python bench.py --sources $S --steps 10 --warmup 2 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('ahead=${lib:-2(shipped)} S=$S ms_per_step %.3f' % d['ms_per_step'])"
    done
  done
done
