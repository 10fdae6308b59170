#!/bin/bash
# Round 4, GPU session 20 (launch replay): timeline of the concurrent step (per-queue busy time, gaps, the tail)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/r04_s20_prof_kt.log 2>&1
cd $R
KT=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/rocpd_timeline.py $KT > gpurun_out/r04_s20_timeline.txt 2>&1
python tools/rocpd_gaps.py $KT > gpurun_out/r04_s20_gaps.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_kt
cat gpurun_out/r04_s20_gaps.txt | head -60
