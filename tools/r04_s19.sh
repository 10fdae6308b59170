#!/bin/bash
# CU-masked side streams: probe + step A/B
cd $GRAFT_REPO_ROOT
timeout 300 python tools/cu_mask_probe.py > gpurun_out/r04_s19_cu_mask_probe.txt 2>&1
timeout 900 python tools/ab_step.py --steps 20 --rounds 3 base w64:wgrad_cus=64 w128:wgrad_cus=128 w192:wgrad_cus=192 \
   w128b128:wgrad_cus=128,branch_cus=128 w64c192:wgrad_cus=64,compute_cus=192 w128c128:wgrad_cus=128,compute_cus=128 > gpurun_out/r04_s19_ab_masks.txt 2>&1
cat gpurun_out/r04_s19_cu_mask_probe.txt gpurun_out/r04_s19_ab_masks.txt
