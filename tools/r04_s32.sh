#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/ab_step.py --steps 20 --rounds 4 old:sw.bn_red_adj=0 new:sw.bn_red_adj=1 > gpurun_out/r04_s32_ab_bn_red_adj.txt 2>&1
timeout 600 python tools/ab_step.py --steps 20 --rounds 2 old_serial:serial=1,sw.bn_red_adj=0 new_serial:serial=1,sw.bn_red_adj=1 >> gpurun_out/r04_s32_ab_bn_red_adj.txt 2>&1
grep -v amdgpu gpurun_out/r04_s32_ab_bn_red_adj.txt
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "bn or batchnorm" 2>&1 | tail -2
