#!/bin/bash
# Round 3, GPU session 1: the whole GPU suite, the interleaved A/B of this round's first switches, kernel stats (serial streams)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/s1_gpu_tests.txt
python tools/ab_step.py --rounds 3 --steps 20 base:RELU_BITS=0,STEM_STATS=0,FUSE_STEM_FWD=0 bits:RELU_BITS=1,STEM_STATS=0,FUSE_STEM_FWD=0 \
   stem:RELU_BITS=1,STEM_STATS=1,FUSE_STEM_FWD=0 stemfwd:RELU_BITS=1,STEM_STATS=1,FUSE_STEM_FWD=1 tail:RELU_BITS=1,STEM_STATS=1,FUSE_STEM_FWD=1,FUSE_STEM_TAIL=1 \
   > gpurun_out/s1_ab.txt 2> gpurun_out/s1_ab.err
cd /tmp && export TMPDIR=/tmp
VIRTEX_AMD_FUSE_STEM_FWD=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/s1_prof_ks.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 80 > gpurun_out/s1_kernel_stats_serial.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_ks
VIRTEX_AMD_FUSE_STEM_FWD=1 python bench.py --no-cpu-baseline --no-fidelity --steps 30 --warmup 10 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
cat gpurun_out/s1_gpu_tests.txt gpurun_out/s1_ab.txt
