#!/bin/bash
# Several chains of sources in flight (C2R_CHAINS=n, csrc/sweep.hip run_chains) against one, alternating on ONE box:
# bench.py --sources S for S = 125 (one GPU's share of the 8-GPU bench), 250, 500 and the 1000-source headline.
# usage: profiles/micro/ab_chains.sh [outfile]
out=${1:-gpurun_out/ab_chains.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
run () {   # $1 = sources, $2 = chains ("" = the library's rule), $3 = steps
  local line
  line=$(C2R_CHAINS=$2 python bench.py --sources $1 --steps $3 --warmup 2 --no-cpu-baseline --no-other-mode --no-small-leg --no-mix-ceiling --no-dropin-leg 2>/dev/null | tail -1)
  echo "S=$1 chains=${2:-rule} $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step=%.3f sum_nbox=%s phih_sum=%.12e" % (d["ms_per_step"], d["check"]["sum_nbox_last_step"], d["check"]["phih_grid_sum"]))')" | tee -a "$out"
}
for rep in 1 2; do
  for c in 1 2 3 4; do run 125 $c 10; done
done
for c in 1 2 3; do run 250 $c 6; done
for c in 1 2 3; do run 500 $c 4; done
for c in 1 2; do run 750 $c 3; done
for c in 1 2; do run 1000 $c 4; done
run 125 "" 10; run 1000 "" 4
