#!/bin/bash
# end of round 6 (LDS planes in the fused first sub-boxes, replayed chains, the plain Gamma bound in gamma_ok): the pass fuzz and the
# whole-step fuzz at HEAD, both modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_tol
timeout 1700 python tests/fuzz_gpu.py 300 70000 fast > gpurun_out/r06_tol/fuzz_fast.txt 2>&1; tail -1 gpurun_out/r06_tol/fuzz_fast.txt
timeout 1700 python tests/fuzz_gpu.py 300 70000 exact > gpurun_out/r06_tol/fuzz_exact.txt 2>&1; tail -1 gpurun_out/r06_tol/fuzz_exact.txt
timeout 1200 python tests/_fuzz_steps.py 100 7000 fast > gpurun_out/r06_tol/step_fuzz_fast.txt 2>&1; tail -1 gpurun_out/r06_tol/step_fuzz_fast.txt
timeout 1200 python tests/_fuzz_steps.py 100 7000 exact > gpurun_out/r06_tol/step_fuzz_exact.txt 2>&1; tail -1 gpurun_out/r06_tol/step_fuzz_exact.txt
