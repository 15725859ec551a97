#!/bin/bash
# Same-box A/B of bench steps: profiles/r04_plane/ab.sh "<label>|ENV=.. ENV=.." ...   (two alternating rounds)
FL="--steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg"
for round in 1 2; do
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  ms=$(env $envs python3 bench.py $FL 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms  nbox %d  phih %.12e' % (d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum']))")
  echo "$label: $ms"
done; done
