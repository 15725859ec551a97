# same-box A/B: how far out the fused sub-box kernel (one workgroup per source) is used when many sources are active
export C2R_SCHED_HINT=0 C2R_SCHED_GUESS=0
for i in 1 2; do for v in "10 64" "15 64" "20 64" "25 64" "20 16" "30 64"; do set -- $v; export C2R_FUSED_QMAX_MANY=$1 C2R_FUSED_MANY=$2
  c=$(python profiles/steps_schedule.py --steps 4 2>/dev/null | python -c "import json,sys; r=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(' '.join('%.4f' % x['wall_s'] for x in r[:4]))")
  d=$(python profiles/steps_schedule.py --steps 3 --sources 100 2>/dev/null | python -c "import json,sys; r=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(' '.join('%.4f' % x['wall_s'] for x in r[:3]))")
  e=$(python bench.py --mesh 64 --sources 20000 --x-init 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(j['ms_per_step'],3))")
  echo "fused q<=$1 when >=$2 active: S=1000 steps 1-4: $c | S=100 steps 1-3: $d | 64^3x20000 cold $e ms"; done; done
