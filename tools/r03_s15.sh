#!/bin/bash
# Round 3, GPU session 15: the flat BatchNorm apply kernels in the adjacent-vector form (tools/probes/stream_probe.hip): per shape
# (tools/bench_bn_apply.py) and at step level
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 400 python tools/bench_bn_apply.py > gpurun_out/s15_bn_apply.txt 2>&1
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 strided:sw.bn_adj=0 adj:sw.bn_adj=1 adj_g4096:sw.bn_adj=1,sw.bn_grid=4096 adj_g16384:sw.bn_adj=1,sw.bn_grid=16384 > gpurun_out/s15_ab.txt 2> gpurun_out/s15_ab.err
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -x -q -m gpu -k "bn or batchnorm" > gpurun_out/s15_tests.txt 2>&1
cat gpurun_out/s15_bn_apply.txt | cut -c1-400; cat gpurun_out/s15_ab.txt; tail -3 gpurun_out/s15_tests.txt; tail -3 gpurun_out/s15_ab.err
