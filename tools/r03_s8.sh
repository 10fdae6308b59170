#!/bin/bash
# Round 3, GPU session 8: epilogue operands requested in front of the K loop (_epipre), tile rule of the statistics epilogues
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=$R/virtex_amd/lib
echo "== epipre" >> gpurun_out/s8_1x1.txt
VIRTEX_AMD_LIB=$L/libvirtex_amd_epipre.so timeout 300 python tools/bench_1x1.py -1,1,6 >> gpurun_out/s8_1x1.txt 2>&1
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 cur_st1:sw.stats_tile=1 cur_st0:sw.stats_tile=0 cur_st2:sw.stats_tile=2 \
  pre_st1:lib=$L/libvirtex_amd_epipre.so,sw.stats_tile=1 pre_st3:lib=$L/libvirtex_amd_epipre.so,sw.stats_tile=3 > gpurun_out/s8_ab.txt 2> gpurun_out/s8_ab.err
VIRTEX_AMD_LIB=$L/libvirtex_amd_epipre.so timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -x -q -m gpu > gpurun_out/s8_tests.txt 2>&1
cat gpurun_out/s8_1x1.txt gpurun_out/s8_ab.txt; tail -3 gpurun_out/s8_tests.txt; tail -3 gpurun_out/s8_ab.err
