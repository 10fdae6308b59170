#!/bin/bash
# joint (tile, slices) plan of the weight gradients: A/B
set -x
R=$GRAFT_REPO_ROOT; cd $R
rm -f gpurun_out/ab10.txt
for v in "VIRTEX_AMD_WGRAD_PLAN=0" "X=1" "VIRTEX_AMD_WGRAD_PLAN=0" "X=1"; do
  env $v VIRTEX_AMD_NT_STORE_MB=200 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab10.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab10.txt
done
cat gpurun_out/ab10.txt
VIRTEX_AMD_WGRAD_PLAN=0 timeout 300 python tools/bench_layers.py > gpurun_out/layers10_old.txt 2>&1
timeout 300 python tools/bench_layers.py > gpurun_out/layers10.txt 2>&1
paste -d'\n' gpurun_out/layers10_old.txt gpurun_out/layers10.txt | grep -v "^/opt" | cut -c88-140 | paste - - | head -40
