#!/bin/bash
# Buffer-descriptor addressing of the DMA kernel: hardware probe, kernel parity on the GPU, A/B against the previous
# library (pointer + zero-page addressing) in one session, non-temporal output stores A/B.
set -x
R=$GRAFT_REPO_ROOT; cd $R
./tools/probes/buf_probe > gpurun_out/buf_probe.txt 2>&1
cat gpurun_out/buf_probe.txt
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/gpu_tests8_kernels.txt
cat gpurun_out/gpu_tests8_kernels.txt
rm -f gpurun_out/ab8.txt
PREV=$R/virtex_amd/lib/libvirtex_amd_prev.so
for v in "VIRTEX_AMD_LIB=$PREV" "X=1" "VIRTEX_AMD_NT_STORE_MB=200" "VIRTEX_AMD_LIB=$PREV" "X=1" "VIRTEX_AMD_NT_STORE_MB=200"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab8.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('${v##*/}', r['ms_per_step'], r['value'])" >> gpurun_out/ab8.txt
done
cat gpurun_out/ab8.txt
timeout 300 python tools/bench_layers.py > gpurun_out/layers8.txt 2>&1
tail -45 gpurun_out/layers8.txt
