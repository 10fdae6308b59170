#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_step.py --steps 20 --rounds 3 base st3:sw.stats_tile=3 st4:sw.stats_tile=4 > gpurun_out/r04_s25_ab_stats_tile.txt 2>&1
timeout 600 python tools/ab_step.py --steps 20 --rounds 2 base_serial:serial=1 st4_serial:serial=1,sw.stats_tile=4 >> gpurun_out/r04_s25_ab_stats_tile.txt 2>&1
grep -v amdgpu gpurun_out/r04_s25_ab_stats_tile.txt
