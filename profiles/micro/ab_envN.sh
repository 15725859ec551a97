# same-box A/B of an environment switch on a few-source bench configuration: ab_envN.sh VAR "v1 v2" MESH SOURCES [steps]
for i in 1 2; do for v in $2; do
  a=$(env $1=$v python bench.py --mesh $3 --sources $4 --steps ${5:-40} --warmup 5 --no-cpu-baseline --no-other-mode $BENCH_ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(j['ms_per_step'],4), 'ms/step  sub-boxes', j['check']['sum_nbox_last_step'])")
  echo "$1=$v: $3^3 x $4 sources $a"; done; done
