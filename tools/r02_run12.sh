#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R
rm -f gpurun_out/ab12.txt
for rep in 1 2; do
for v in "VIRTEX_AMD_WGRAD_PLAN=0" "VIRTEX_AMD_WGRAD_PLAN=2" "VIRTEX_AMD_WGRAD_PLAN=1 VIRTEX_AMD_WGRAD_PLAN_MAXK=13000" "VIRTEX_AMD_WGRAD_PLAN=3 VIRTEX_AMD_WGRAD_PLAN_MAXK=13000" "VIRTEX_AMD_WGRAD_PLAN=3 VIRTEX_AMD_WGRAD_PLAN_MAXK=8000" "VIRTEX_AMD_WGRAD_PLAN=3 VIRTEX_AMD_WGRAD_STREAM=0" "VIRTEX_AMD_WGRAD_PLAN=0 VIRTEX_AMD_WGRAD_STREAM=0"; do
  env $v VIRTEX_AMD_NT_STORE_MB=200 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab12.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab12.txt
done; done
cat gpurun_out/ab12.txt; tail -3 gpurun_out/ab12.err
