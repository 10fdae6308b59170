#!/bin/bash
# Round 5, GPU session 17: the reference's module graph through stock PyTorch-ROCm on the same GPU (secondary comparison)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 420 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --stock-pytorch-baseline --steps 20 --warmup 10 > gpurun_out/r05_s17_stock.json 2> gpurun_out/r05_s17_stock.err
echo rc=$?
tail -c 1500 gpurun_out/r05_s17_stock.json; grep -v amdgpu.ids gpurun_out/r05_s17_stock.err | tail -5
