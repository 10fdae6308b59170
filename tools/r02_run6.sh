#!/bin/bash
# host-lead diagnosis: is the 2 ms compute-queue gap at the text-head turn host time or a dependency?
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 400 python tools/host_trace.py > gpurun_out/host_trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_kt6 -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 6 --warmup 3 > $R/gpurun_out/prof_kt6.log 2>&1
cd $R
python tools/rocpd_timeline.py $(find gpurun_out/prof_kt6 -name "*.db" | head -1) 9.8 16.0 > gpurun_out/timeline_turn.txt 2>&1
find gpurun_out -name "*.db" -delete
cat gpurun_out/host_trace.txt | tail -60
