#!/bin/bash
# Round 5, GPU session 12: re-sweep of the side-stream switches now that the stage-1 conv3 weight gradients left the side stream
set -x
cd $GRAFT_REPO_ROOT
timeout 600 python tools/ab_step.py --steps 20 --rounds 3 base mc400:sw.gen3_mc=400 mc1500:sw.gen3_mc=1500 mc0:sw.gen3_mc=0 sk384:sw.splitk_blocks=384 sk768:sw.splitk_blocks=768 > gpurun_out/r05_s12_ab_side.txt 2>&1
timeout 600 python tools/ab_step.py --steps 20 --rounds 3 base st0:sw.stats_tile=0 st5:sw.stats_tile=5 gen3_60:sw.gen3=60 gen3_100:sw.gen3=100 eff84:sw.mc_eff128=84 > gpurun_out/r05_s12_ab_tiles.txt 2>&1
cat gpurun_out/r05_s12_ab_side.txt gpurun_out/r05_s12_ab_tiles.txt
