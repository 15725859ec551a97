cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/crash
for i in 1; do
  python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/crash/full_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/crash/full_$i.log | cut -c1-120)"
  if grep -q "Fatal Python error\|core dumped\|Aborted" gpurun_out/crash/full_$i.log; then grep -n "Fatal Python error" -B5 -A40 gpurun_out/crash/full_$i.log | head -120; break; fi
done
