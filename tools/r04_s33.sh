#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/ab_step.py --steps 20 --rounds 3 base noexpand:sw.expand1x1=0 noshared:sw.conv3x3_shared=0 nowg3:sw.wgrad3x3=0 nostem:sw.stem_stream=0 mc100:sw.gen3_mc=100 mc400:sw.gen3_mc=400 > gpurun_out/r04_s33_ab_revalidate.txt 2>&1
grep -v amdgpu gpurun_out/r04_s33_ab_revalidate.txt
