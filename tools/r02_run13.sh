#!/bin/bash
# kernel traces of the current step (three streams and serial) + default bench line
set -x
R=$GRAFT_REPO_ROOT; cd $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/prof_ks.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 70 > gpurun_out/kernel_stats_serial.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_kt -name "*.db" | head -1) 70 > gpurun_out/kernel_stats.txt
find gpurun_out -name "*.db" -delete
rm -rf gpurun_out/prof_ks gpurun_out/prof_kt
head -45 gpurun_out/kernel_stats_serial.txt | cut -c1-200
