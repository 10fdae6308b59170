#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/replay_dump.py --batch 8 > gpurun_out/r05_s11_replay_ops.txt 2>&1
grep -c "" gpurun_out/r05_s11_replay_ops.txt
grep " aten:" gpurun_out/r05_s11_replay_ops.txt | awk '{print $2}' | sort | uniq -c | sort -rn
