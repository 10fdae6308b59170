#!/bin/bash
# Round 3, GPU session 31: kernel traces at HEAD (with the shared-tile 3x3 kernel in the step)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/s31_prof_kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/s31_prof_ks.log 2>&1
cd $R
KT=$(find gpurun_out/prof_kt -name "*.db" | head -1); KS=$(find gpurun_out/prof_ks -name "*.db" | head -1)
python tools/rocpd_stats.py $KT 80 > gpurun_out/s31_kernel_stats.txt
python tools/rocpd_stats.py $KS 80 > gpurun_out/s31_kernel_stats_serial.txt
python tools/rocpd_timeline.py $KT > gpurun_out/s31_timeline.txt 2>&1
python tools/rocpd_gaps.py $KT > gpurun_out/s31_gaps.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_kt gpurun_out/prof_ks
head -12 gpurun_out/s31_kernel_stats_serial.txt; grep -i "conv3x3_shared" gpurun_out/s31_kernel_stats_serial.txt
