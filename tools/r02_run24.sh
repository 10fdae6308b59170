#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R
VIRTEX_AMD_EXPAND1X1=1 timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "expand1x1" 2>&1 | tail -3 > gpurun_out/gpu_tests24.txt
cat gpurun_out/gpu_tests24.txt
python tools/bench_1x1.py -1 2>&1 | grep "fwd+stats" > gpurun_out/bench_1x1_e0.txt
VIRTEX_AMD_EXPAND1X1=1 python tools/bench_1x1.py -1 2>&1 | grep "fwd+stats" > gpurun_out/bench_1x1_e1.txt
paste -d'\n' gpurun_out/bench_1x1_e0.txt gpurun_out/bench_1x1_e1.txt
rm -f gpurun_out/ab24.txt
for rep in 1 2 3; do
for v in "VIRTEX_AMD_EXPAND1X1=0" "VIRTEX_AMD_EXPAND1X1=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab24.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab24.txt
done; done
cat gpurun_out/ab24.txt; tail -2 gpurun_out/ab24.err
