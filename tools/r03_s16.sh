#!/bin/bash
# Round 3, GPU session 16: block -> tile order of the contraction kernel (XCD-contiguous ranges vs plain) on the HBM-bound layers and the step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for v in 0 1; do echo "== tile_order $v" >> gpurun_out/s16_1x1.txt; VIRTEX_AMD_KFLAGS=$((v*32)) timeout 300 python tools/bench_1x1.py -1 >> gpurun_out/s16_1x1.txt 2>&1; done
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 xcd:sw.tile_order=0 plain:sw.tile_order=1 > gpurun_out/s16_ab.txt 2> gpurun_out/s16_ab.err
cat gpurun_out/s16_1x1.txt gpurun_out/s16_ab.txt; tail -3 gpurun_out/s16_ab.err
