#!/bin/bash
# Round 5, GPU session 1: the fused conv3 backward (csrc/conv3_bwd.hip) -- GPU tests, per-shape timing against the three
# launches it replaces, step-level A/B, and a default bench line (launch replay with the out= / refill forms).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py tests/test_model_parity.py -x -q -m gpu -k "conv3_backward_in_one or fused_conv3_backward or batchnorm_backward_fused_in_gemm" 2>&1 | tail -5 > gpurun_out/r05_s1_tests.txt
timeout 200 python tools/bench_conv3_bwd.py > gpurun_out/r05_s1_conv3_bwd.txt 2>&1
timeout 300 python tools/ab_step.py --steps 20 --rounds 3 fused off:FUSE_CONV3_BWD=0 > gpurun_out/r05_s1_ab.txt 2>&1
timeout 300 python tools/ab_step.py --steps 20 --rounds 2 fused:serial=1 off:FUSE_CONV3_BWD=0,serial=1 > gpurun_out/r05_s1_ab_serial.txt 2>&1
timeout 300 python -m pytest tests/test_replay.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r05_s1_replay_tests.txt
timeout 400 python bench.py --steps 20 --warmup 10 > gpurun_out/r05_s1_bench.json 2> gpurun_out/r05_s1_bench.err
cat gpurun_out/r05_s1_tests.txt gpurun_out/r05_s1_conv3_bwd.txt gpurun_out/r05_s1_ab.txt gpurun_out/r05_s1_ab_serial.txt gpurun_out/r05_s1_replay_tests.txt
tail -c 600 gpurun_out/r05_s1_bench.err
