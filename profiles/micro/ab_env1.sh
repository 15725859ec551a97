# same-box A/B of an environment switch on BASELINE configs[1] (128^3, 1 source): ab_env1.sh VAR "v1 v2 ..."
for i in 1 2 3; do for v in $2; do
  a=$(env $1=$v python bench.py --mesh 128 --sources 1 --steps 200 --warmup 10 --no-cpu-baseline --no-other-mode $BENCH_ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(j['ms_per_step'],4), 'ms/step', j['check'])")
  echo "$1=$v: 128^3 x 1 source $a"; done; done
