#!/bin/bash
# Round 4, GPU session 8: generation-3 weight-gradient kernels -- hardware tests, per-shape table with race screen
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "generation3" -x 2>&1 | tail -4 > gpurun_out/r04_s8_tests.txt
cat gpurun_out/r04_s8_tests.txt
timeout 900 python tools/bench_gen3_mc.py --race > gpurun_out/r04_s8_gen3_mc.txt 2>&1
cat gpurun_out/r04_s8_gen3_mc.txt
