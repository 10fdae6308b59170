#!/bin/bash
# Round 3, GPU session 27: 3x3 / stride-1 convolutions (forward, input gradient) with the A tile shared by the three taps of a
# filter row (conv3x3_kernel.h) against the generic implicit-GEMM kernel: per layer and at step level
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -x -q -m gpu -k "conv" > gpurun_out/s27_tests.txt 2>&1
for v in 0 1; do echo "== conv3x3_shared $v" >> gpurun_out/s27_layers.txt; VIRTEX_AMD_CONV3X3_SHARED=$v timeout 300 python tools/bench_layers.py 2>&1 | grep -i "3x3\|conv2" >> gpurun_out/s27_layers.txt; done
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 generic:sw.conv3x3_shared=0 shared:sw.conv3x3_shared=1 > gpurun_out/s27_ab.txt 2> gpurun_out/s27_ab.err
tail -3 gpurun_out/s27_tests.txt; cat gpurun_out/s27_layers.txt | cut -c1-220; cat gpurun_out/s27_ab.txt; tail -3 gpurun_out/s27_ab.err
