#!/bin/bash
# Round 5, GPU session 8: timeline of one replayed step (per-queue busy time, compute-queue gaps), kernel stats of the three-stream step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/r05_s8_prof_kt.log 2>&1
cd $R
KT=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/rocpd_timeline.py $KT > gpurun_out/r05_s8_timeline.txt 2>&1
python tools/rocpd_gaps.py $KT > gpurun_out/r05_s8_gaps.txt 2>&1
python tools/rocpd_stats.py $KT 45 > gpurun_out/r05_s8_kernel_stats.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_kt
cat gpurun_out/r05_s8_timeline.txt | head -50
