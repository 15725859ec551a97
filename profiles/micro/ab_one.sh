# same-box A/B of library variants on BASELINE configs[1] (128^3, 1 source): ms per outer iteration, graph replay on
#   usage: ab_one.sh <variant tags...>     (libraries c2-ray3dm_amd/libc2ray_hip_<tag>.so; "base" is always included)
for i in 1 2 3; do for v in base "$@"; do if [ $v = base ]; then unset C2RAY_HIP_LIB; else export C2RAY_HIP_LIB=$PWD/c2-ray3dm_amd/libc2ray_hip_$v.so; fi
  a=$(python bench.py --mesh 128 --sources 1 --steps 200 --warmup 10 --no-cpu-baseline --no-other-mode $BENCH_ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(j['ms_per_step'],4), 'ms/step, sub-boxes', j['config'].get('mean_subboxes_per_source'))")
  echo "$v: 128^3 x 1 source $a"; done; done
