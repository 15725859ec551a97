# same-box A/B in the launch-bound regime: 128^3 x 1 source (BASELINE configs[1]), 256^3 x 1000 cold, step 1-2 of the schedule
for i in 1 2; do for v in base "$@"; do if [ $v = base ]; then unset C2RAY_HIP_LIB; else export C2RAY_HIP_LIB=$PWD/c2-ray3dm_amd/libc2ray_hip_$v.so; fi
  a=$(python bench.py --mesh 128 --sources 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(j['ms_per_step'],3))")
  b=$(python bench.py --mesh 256 --sources 1000 --x-init 2e-4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(j['ms_per_step'],3))")
  c=$(python profiles/steps_schedule.py --steps 2 2>/dev/null | python -c "import json,sys; r=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(' '.join('%.4f s/%d it' % (x['wall_s'], x['outer_iterations']) for x in r[:2]))")
  d=$(python profiles/steps_schedule.py --steps 2 --sources 100 2>/dev/null | python -c "import json,sys; r=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(' '.join('%.4f s/%d it' % (x['wall_s'], x['outer_iterations']) for x in r[:2]))")
  echo "$v: 128^3x1 $a ms/step | 256^3x1000 cold $b ms/step | schedule S=1000 steps 1,2: $c | S=100: $d"; done; done
