#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/ab_step.py --steps 20 --rounds 3 g80:sw.gen3=80 g60:sw.gen3=60 g70:sw.gen3=70 g90:sw.gen3=90 g100:sw.gen3=100 > gpurun_out/r04_s44_ab_gen3_thr.txt 2>&1
grep -v amdgpu gpurun_out/r04_s44_ab_gen3_thr.txt
