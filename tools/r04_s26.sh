#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/ab_step.py --steps 20 --rounds 3 st0:sw.stats_tile=0 st4:sw.stats_tile=4 st5:sw.stats_tile=5 st6:sw.stats_tile=6 st7:sw.stats_tile=7 > gpurun_out/r04_s26_ab_stats_tile.txt 2>&1
grep -v amdgpu gpurun_out/r04_s26_ab_stats_tile.txt
