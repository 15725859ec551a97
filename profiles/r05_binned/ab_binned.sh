#!/bin/bash
# Same-box A/B of the source-binned far shells (csrc/kernels_binned.hpp) against the source-major launches, 256^3 x 1000 sources:
# isothermal with atomics (the headline), non-isothermal (two atomics per visit), ordered rates (per-source grids + ordered sum),
# ordered rates non-isothermal.  C2R_BINNED = 0 never / 1 wherever the coverage rule holds.
run () { env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-28s %-28s ms_per_step %8.2f  sum_nbox %d  phih_sum %.12e' % ('$1', '$2', d['ms_per_step'], d['check']['sum_nbox_last_step'], d['check']['phih_grid_sum']))"; }
for rep in 1 2; do
  for flags in "" "--thermal" "--deterministic" "--deterministic --thermal"; do
    run C2R_BINNED=0 "$flags"
    run C2R_BINNED=1 "$flags"
  done
done
