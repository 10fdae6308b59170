#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R
rm -f gpurun_out/ab21.txt
for rep in 1 2; do
for v in "X=1" "VIRTEX_AMD_EPI_NT_LOADS=1" "VIRTEX_AMD_KFLAGS=32" "VIRTEX_AMD_BN_UNROLL=4"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab21.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab21.txt
done; done
cat gpurun_out/ab21.txt
