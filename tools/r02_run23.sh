#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/gpu_tests23.txt
cat gpurun_out/gpu_tests23.txt
rm -f gpurun_out/ab23.txt
PREV=$R/virtex_amd/lib/libvirtex_amd_prev6.so
for rep in 1 2 3; do
for v in "VIRTEX_AMD_LIB=$PREV" "X=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab23.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('${v##*/}', r['ms_per_step'], r['value'])" >> gpurun_out/ab23.txt
done; done
cat gpurun_out/ab23.txt

