#!/bin/bash
cd $GRAFT_REPO_ROOT
(echo "== config 4: bs 128, H 1024, F 4096"; timeout 300 python tools/sweep_text_tiles.py 128 1024 4096; echo "== config 5: bs 64, H 2048, F 8192"; timeout 300 python tools/sweep_text_tiles.py 64 2048 8192) > gpurun_out/r05_s13_text_tiles.txt 2>&1
grep -v amdgpu gpurun_out/r05_s13_text_tiles.txt
