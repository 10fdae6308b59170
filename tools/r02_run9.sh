#!/bin/bash
# split-K block order A/B (slice-major XCD ranges vs tile-major), tile sweep on the buffer-addressed kernel
set -x
R=$GRAFT_REPO_ROOT; cd $R
rm -f gpurun_out/ab9.txt
for v in "VIRTEX_AMD_KFLAGS=16" "X=1" "VIRTEX_AMD_KFLAGS=16" "X=1"; do
  env $v VIRTEX_AMD_NT_STORE_MB=200 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab9.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab9.txt
done
cat gpurun_out/ab9.txt
VIRTEX_AMD_KFLAGS=16 timeout 300 python tools/bench_layers.py > gpurun_out/layers9_tilemajor.txt 2>&1
timeout 300 python tools/bench_layers.py > gpurun_out/layers9.txt 2>&1
paste -d'\n' gpurun_out/layers9_tilemajor.txt gpurun_out/layers9.txt | grep -v "^/opt" | cut -c1-140
timeout 600 python tools/sweep_tiles.py -1,0,1,2,3,4,5,11,12 > gpurun_out/sweep9.txt 2>&1
grep -c . gpurun_out/sweep9.txt; grep "auto loses" gpurun_out/sweep9.txt | cut -c1-250
