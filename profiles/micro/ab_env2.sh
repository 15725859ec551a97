# alternating same-box runs under settings of two environment variables: ab_env2.sh "A=1 B=2" "A=3" ...  ("" = none)
for i in 1 2; do for v in "" "$@"; do env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('[$v]', round(j['ms_per_step'],2))"; done; done
