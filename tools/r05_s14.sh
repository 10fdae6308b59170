#!/bin/bash
# Round 5, GPU session 14: run-to-run determinism with the fused conv3 backward (a stream race would show), sustained load
cd $GRAFT_REPO_ROOT
timeout 300 python tools/determinism.py > gpurun_out/r05_s14_determinism.txt 2>&1
true
grep -v amdgpu gpurun_out/r05_s14_determinism.txt | tail -8; grep -v amdgpu gpurun_out/r05_s14_sustained.txt | tail -3
