# alternating same-box runs of the default library under different environment settings
# usage: ab_env.sh VAR val1 val2 ...   (an empty value = unset)
VAR=$1; shift
for i in 1 2 3; do for v in "" "$@"; do if [ -z "$v" ]; then unset $VAR; else export $VAR=$v; fi; python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$VAR=$v', round(j['ms_per_step'],2))"; done; done
