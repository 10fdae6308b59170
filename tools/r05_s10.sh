#!/bin/bash
# Round 5, GPU session 10: the whole library without hipcc's SLP vectoriser (v_pk_*_f32 from scalar fp32 code) against the default build
set -x
cd $GRAFT_REPO_ROOT
timeout 400 python tools/ab_step.py --steps 20 --rounds 3 base noslp:lib=libvirtex_amd_noslp.so > gpurun_out/r05_s10_ab.txt 2>&1
timeout 300 python tools/ab_step.py --steps 20 --rounds 2 base:serial=1 noslp:lib=libvirtex_amd_noslp.so,serial=1 > gpurun_out/r05_s10_ab_serial.txt 2>&1
cat gpurun_out/r05_s10_ab.txt gpurun_out/r05_s10_ab_serial.txt
