#!/bin/bash
# from which shell the plane-ordered mapping is used (C2R_XCD_QMIN; shipped: 16), 256^3 x 1000 sources
run () { env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-20s ms_per_step %8.2f  sum_nbox %d' % ('$1', d['ms_per_step'], d['check']['sum_nbox_last_step']))"; }
for rep in 1 2; do for q in 6 16 32 48; do run C2R_XCD_QMIN=$q; done; done
