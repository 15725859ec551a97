#!/bin/bash
# the 14-step schedule at the end of round 5 (plane-ordered mapping, 8 waves), both sweep modes, library defaults
mkdir -p gpurun_out/r05_xcd
for m in 1 0; do
  C2R_SWEEP_MODE=$m python profiles/steps_schedule.py > gpurun_out/r05_xcd/schedule_head_mode$m.jsonl 2>/dev/null
  tail -1 gpurun_out/r05_xcd/schedule_head_mode$m.jsonl
done
