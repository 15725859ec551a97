# same-box A/B of the sweep schedule switches (C2R_SCHED_HINT: run ahead to where the previous pass ended; C2R_SCHED_GUESS: grid.z from the previous pass's counts)
for i in 1 2; do for v in "1 1" "0 0" "1 0" "0 1"; do set -- $v; export C2R_SCHED_HINT=$1 C2R_SCHED_GUESS=$2
  a=$(python bench.py --mesh 128 --sources 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(j['ms_per_step'],3))")
  c=$(python profiles/steps_schedule.py --steps 3 2>/dev/null | python -c "import json,sys; r=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(' '.join('%.4f' % x['wall_s'] for x in r[:3]))")
  d=$(python profiles/steps_schedule.py --steps 3 --sources 100 2>/dev/null | python -c "import json,sys; r=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(' '.join('%.4f' % x['wall_s'] for x in r[:3]))")
  echo "hint=$1 guess=$2: 128^3x1 $a ms/step | S=1000 steps 1-3: $c | S=100 steps 1-3: $d"; done; done
