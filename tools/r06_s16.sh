#!/bin/bash
# Round 6, GPU session 16: 6 000 replays of the recorded step on the final build (time per block, loss, device / host memory).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/sustained_replay.py 6000 > gpurun_out/r06_s16_sustained_replay.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06_s16_sustained_replay.txt | tail -6
