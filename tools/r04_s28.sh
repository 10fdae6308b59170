#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/ab_step.py --steps 20 --rounds 4 old:sw.gen3_s2=1 new:sw.gen3_s2=0 > gpurun_out/r04_s28_ab_gen3_s2.txt 2>&1
grep -v amdgpu gpurun_out/r04_s28_ab_gen3_s2.txt
