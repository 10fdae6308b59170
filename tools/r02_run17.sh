#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R
rm -f gpurun_out/ab17.txt
for rep in 1 2; do
for v in "VIRTEX_AMD_SPLITK_BLOCKS=512" "VIRTEX_AMD_SPLITK_BLOCKS=384" "VIRTEX_AMD_SPLITK_BLOCKS=768" "VIRTEX_AMD_SPLITK_BLOCKS=1024" "VIRTEX_AMD_SPLITK_BLOCKS=256"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab17.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('${v##*/}', r['ms_per_step'], r['value'])" >> gpurun_out/ab17.txt
done; done
cat gpurun_out/ab17.txt
