cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
FL="--sources 125 --steps 4 --warmup 2 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
rm -rf /tmp/kt1
C2R_BENCH_PROFILE=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o kt -- python3 bench.py $FL --option chains=1 --option chain_graph=0 > gpurun_out/kt1_bench.json 2> gpurun_out/kt1.err
T=$(find /tmp/kt1 -name '*kernel_trace.csv' | head -1)
python3 profiles/per_shell_efficiency.py $T 125 256 8.35 > gpurun_out/per_shell_125.txt
cat gpurun_out/per_shell_125.txt
FL="--steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-mix-ceiling --no-dropin-leg --no-small-leg --no-configs-leg"
rm -rf /tmp/kt2
C2R_BENCH_PROFILE=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o kt -- python3 bench.py $FL --option xcd_order=0 > gpurun_out/kt2_bench.json 2> gpurun_out/kt2.err
T=$(find /tmp/kt2 -name '*kernel_trace.csv' | head -1)
python3 profiles/per_shell_efficiency.py $T 1000 256 8.35 > gpurun_out/per_shell_1000_plain.txt
cat gpurun_out/per_shell_1000_plain.txt
