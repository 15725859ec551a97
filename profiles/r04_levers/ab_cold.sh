#!/bin/bash
# cold 256^3 x 1000 (x = 2e-4: every source retires inside the fused first sub-boxes): ms per outer iteration, alternating
FL="--x-init 2e-4 --steps 40 --warmup 5 --no-cpu-baseline --no-other-mode --no-small-leg"
for round in 1 2; do
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  ms=$(env $envs C2R_BENCH_PROFILE=0 python3 bench.py $FL 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  nbox %d  xh %.12e' % (d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['xh_intermed_sum']))")
  echo "$label: $ms"
done; done
