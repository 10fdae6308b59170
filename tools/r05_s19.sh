#!/bin/bash
# Round 5, GPU session 19: kernel trace of the reference's module graph through stock PyTorch-ROCm (channels_last, the faster layout):
# where the 59 ms of the stock step go, kernel by kernel, next to profiles/r05_kernel_stats_serial.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
VTX_STOCK_LAYOUT=channels_last timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stock -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --launch eager --steps 1 --warmup 1 --stock-pytorch-baseline > $R/gpurun_out/r05_s19_stock.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_stock -name "*.db" | head -1) 60 --tail-ms 500 > gpurun_out/r05_s19_kernel_stats_stock.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_stock
head -45 gpurun_out/r05_s19_kernel_stats_stock.txt | cut -c1-170
grep -h '^{' gpurun_out/r05_s19_stock.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('stock_pytorch_baseline'))[:500])"
