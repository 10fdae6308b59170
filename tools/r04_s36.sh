#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1700 python tools/ab_step.py --steps 20 --rounds 4 eff84:sw.mc_eff128=84 eff76:sw.mc_eff128=76 eff70:sw.mc_eff128=70 eff60:sw.mc_eff128=60 eff50:sw.mc_eff128=50 > gpurun_out/r04_s36_ab_mc_eff.txt 2>&1
grep -v amdgpu gpurun_out/r04_s36_ab_mc_eff.txt
