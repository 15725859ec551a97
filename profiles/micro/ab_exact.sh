# the EXACT sweep mode at the bench workload, two runs (round 4: its rates come from the routine both modes share, kernels_common.hpp rates_fast --
# 190.8 -> 176 ms per step; before that the same change lived here as a patch measured at 195.5 -> 183.5 ms)
for i in 1 2; do
  python bench.py --sweep-mode exact --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('exact mode', round(j['ms_per_step'],2), 'ms/step', j['check']['sum_nbox_last_step'], j['check']['phih_grid_sum'])"
done
