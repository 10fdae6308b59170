#!/bin/bash
# Round 6, GPU session 7: step-level A/B of the two switch settings the per-shape probe of the k-major weight gradients liked
# (tools/r06_s6.sh: gen3_mc = 100 is 13-16 % faster per launch at 14x14 / 7x7, splitk_blocks = 256 5-9 % faster at 56x56).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/ab_step.py --steps 20 --rounds 3 base g100:sw.gen3_mc=100 g200:sw.gen3_mc=200 g400:sw.gen3_mc=400 sk256:sw.splitk_blocks=256 sk384:sw.splitk_blocks=384 > gpurun_out/r06_s7_ab.txt 2>&1
timeout 600 python tools/ab_step.py --steps 20 --rounds 2 base:serial=1 g100:sw.gen3_mc=100,serial=1 sk256:sw.splitk_blocks=256,serial=1 > gpurun_out/r06_s7_ab_serial.txt 2>&1
tail -7 gpurun_out/r06_s7_ab.txt; tail -4 gpurun_out/r06_s7_ab_serial.txt
