#!/bin/bash
# Round 4, GPU session 7: where the time is now -- kernel trace of the serial-stream step at HEAD, default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/r04_s7_prof_ks.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 90 > gpurun_out/r04_s7_kernel_stats_serial.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_ks
python bench.py --no-cpu-baseline --no-fidelity > gpurun_out/r04_s7_bench.json 2> gpurun_out/r04_s7_bench.err
head -75 gpurun_out/r04_s7_kernel_stats_serial.txt
