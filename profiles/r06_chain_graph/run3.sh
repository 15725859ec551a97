cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_share125; export TMPDIR=/tmp
FL="--sources 125 --steps 10 --warmup 5 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
for cg in 1 0; do
  rm -rf /tmp/kt$cg
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$cg -o kt -- python3 bench.py $FL --option chain_graph=$cg > gpurun_out/r06_share125/kt_bench_cg$cg.json 2> gpurun_out/r06_share125/kt_cg$cg.err
  T=$(find /tmp/kt$cg -name '*kernel_trace.csv' | head -1)
  python3 profiles/timeline_union.py $T 0.625 10 > gpurun_out/r06_share125/timeline_cg$cg.txt
done
head -50 gpurun_out/r06_share125/timeline_cg1.txt; head -3 gpurun_out/r06_share125/timeline_cg0.txt
