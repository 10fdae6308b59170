#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1700 python tools/ab_step.py --steps 20 --rounds 3 base skb256:sw.splitk_blocks=256 skb1024:sw.splitk_blocks=1024 eff70:sw.mc_eff128=70 eff100:sw.mc_eff128=100 grid4k:sw.bn_grid=4096 grid16k:sw.bn_grid=16384 headord0:models.HEAD_ORDER_BRANCH_FIRST=0 splitprep:models.SPLIT_WEIGHT_PREP=1 > gpurun_out/r04_s35_ab_sweeps.txt 2>&1
grep -v amdgpu gpurun_out/r04_s35_ab_sweeps.txt
