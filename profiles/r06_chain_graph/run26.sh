cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/crash
for i in 1 2 3 4 5 6; do
  python -X faulthandler -m pytest tests/test_gpu_chains.py tests/test_gpu_few_sources.py tests/test_gpu_xcd_order.py tests/test_gpu_allfrac.py tests/test_gpu_fuzz.py -q -m gpu -p no:randomly > gpurun_out/crash/stress_$i.log 2>&1
  echo "stress $i rc=$? $(tail -1 gpurun_out/crash/stress_$i.log | cut -c1-100)"
  if grep -q "Fatal Python error\|core dumped\|Aborted" gpurun_out/crash/stress_$i.log; then grep -n "Fatal Python error" -B5 -A40 gpurun_out/crash/stress_$i.log | head -100; break; fi
done
python -X faulthandler tests/fuzz_chains_gpu.py 200 300 1 > gpurun_out/crash/fz.log 2>&1; echo "fuzz rc=$? $(tail -1 gpurun_out/crash/fz.log | cut -c1-200)"
