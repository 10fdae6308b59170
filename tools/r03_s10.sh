#!/bin/bash
# Round 3, GPU session 10: the fused stem backward tail on 2x2 pixel quads (PoolQuad) at step level; kernel times of the tail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 tail0 tail1:FUSE_STEM_TAIL=1 tail1_st0:FUSE_STEM_TAIL=1,sw.stats_tile=0 tail0_st0:sw.stats_tile=0 > gpurun_out/s10_ab.txt 2> gpurun_out/s10_ab.err
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "pool or stem or bn or batchnorm" > gpurun_out/s10_tests.txt 2>&1
VIRTEX_AMD_FUSE_STEM_TAIL=1 timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu > gpurun_out/s10_tests_tail.txt 2>&1
cd /tmp && export TMPDIR=/tmp
VIRTEX_AMD_FUSE_STEM_TAIL=1 VIRTEX_AMD_STATS_TILE=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/s10_prof_ks.log 2>&1
cd $R
KS=$(find gpurun_out/prof_ks -name "*.db" | head -1)
python tools/rocpd_stats.py $KS 70 > gpurun_out/s10_kernel_stats_serial.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_ks
cat gpurun_out/s10_ab.txt; tail -3 gpurun_out/s10_tests.txt; tail -3 gpurun_out/s10_tests_tail.txt; tail -3 gpurun_out/s10_ab.err
grep -i "pool\|bn_bwd\|bn_reduce\|stem" gpurun_out/s10_kernel_stats_serial.txt | head -20
