#!/bin/bash
# Round 3, GPU session 7: the fused BatchNorm-backward epilogue with its operand loads hoisted in front of each 16-row step
# (libvirtex_amd.so: 4 waves/SIMD, all steps' operands up front; _epi40: per step; _epi0: the build before the change)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=virtex_amd/lib
for v in epi0 cur epi40; do
  lib=$L/libvirtex_amd_$v.so; [ $v = cur ] && lib=$L/libvirtex_amd.so
  echo "== $v" >> gpurun_out/s7_1x1.txt
  VIRTEX_AMD_LIB=$R/$lib timeout 300 python tools/bench_1x1.py -1,1,6 >> gpurun_out/s7_1x1.txt 2>&1
done
timeout 600 python tools/ab_step.py --rounds 3 --steps 20 epi0:lib=$R/$L/libvirtex_amd_epi0.so cur:lib=$R/$L/libvirtex_amd.so epi40:lib=$R/$L/libvirtex_amd_epi40.so > gpurun_out/s7_ab.txt 2> gpurun_out/s7_ab.err
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py tests/test_model_parity.py -x -q -m gpu > gpurun_out/s7_tests.txt 2>&1
cat gpurun_out/s7_1x1.txt gpurun_out/s7_ab.txt; tail -3 gpurun_out/s7_tests.txt
