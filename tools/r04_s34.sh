#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/ab_step.py --steps 20 --rounds 4 mc200:sw.gen3_mc=200 mc0:sw.gen3_mc=0 mc400:sw.gen3_mc=400 mc800:sw.gen3_mc=800 mc1500:sw.gen3_mc=1500 > gpurun_out/r04_s34_ab_gen3_mc.txt 2>&1
grep -v amdgpu gpurun_out/r04_s34_ab_gen3_mc.txt
