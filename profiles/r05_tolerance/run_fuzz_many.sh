#!/bin/bash
# late round 5: the pass fuzz with the many-source draws (64 - 300 sources: chains; 769 - 900: the plane-ordered block mapping)
mkdir -p gpurun_out/r05x
python -m pytest tests/test_gpu_fuzz.py -x -q -k random_pass > gpurun_out/r05x/pytest_fuzz.log 2>&1; tail -2 gpurun_out/r05x/pytest_fuzz.log
python tests/fuzz_gpu.py 300 60000 fast > gpurun_out/r05x/fuzz_many_fast.txt 2>&1; tail -1 gpurun_out/r05x/fuzz_many_fast.txt
python tests/fuzz_gpu.py 300 60000 exact > gpurun_out/r05x/fuzz_many_exact.txt 2>&1; tail -1 gpurun_out/r05x/fuzz_many_exact.txt
