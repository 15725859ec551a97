cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cg in 0 1 0 1; do
  python profiles/steps_schedule.py --sources 1000 --share-of 8 --option chain_graph=$cg > gpurun_out/sched125_cg${cg}_$RANDOM.jsonl 2>/dev/null
done
for f in gpurun_out/sched125_cg*.jsonl; do echo $f; tail -1 $f; python - $f <<'PY'
import sys, json
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"step"')]
print([ (r["step"], r["outer_iterations"], round(r["wall_s"],3)) for r in rows][:14]); print(rows[-1]["info"])
PY
done
