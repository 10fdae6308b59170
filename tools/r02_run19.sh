#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R


rm -f gpurun_out/ab19.txt
for rep in 1 2; do
for v in "VIRTEX_AMD_MC_STAGES=3" "VIRTEX_AMD_MC_STAGES=2"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab19.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab19.txt
done; done
cat gpurun_out/ab19.txt
