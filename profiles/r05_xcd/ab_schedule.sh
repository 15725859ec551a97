#!/bin/bash
# The 14-step schedule (profiles/steps_schedule.py, 256^3 x 1000 sources from the cold start, fast mode) with and without the
# plane-ordered mapping on one box.
for m in 0 1; do
  C2R_SWEEP_MODE=1 C2R_XCD_ORDER=$m python profiles/steps_schedule.py > gpurun_out/r05_xcd/schedule_xcd$m.jsonl 2>/dev/null
  tail -1 gpurun_out/r05_xcd/schedule_xcd$m.jsonl
done
