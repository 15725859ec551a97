VAR=$1; shift
for i in 1 2 3; do for v in "" "$@"; do if [ -z "$v" ]; then unset $VAR; else export $VAR=$v; fi; python bench.py --sources 125 --steps 10 --warmup 1 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$VAR=$v', round(j['ms_per_step'],3))"; done; done
