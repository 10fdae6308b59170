#!/bin/bash
# Round 3, GPU session 21: evidence refresh at HEAD after sessions 18-20 (GPU suite, default bench, kernel traces)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/s21_gpu_tests.txt
python bench.py > gpurun_out/s21_bench_default.json 2> gpurun_out/s21_bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/s21_prof_kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/s21_prof_ks.log 2>&1
cd $R
KT=$(find gpurun_out/prof_kt -name "*.db" | head -1); KS=$(find gpurun_out/prof_ks -name "*.db" | head -1)
python tools/rocpd_stats.py $KT 80 > gpurun_out/s21_kernel_stats.txt
python tools/rocpd_stats.py $KS 80 > gpurun_out/s21_kernel_stats_serial.txt
python tools/rocpd_timeline.py $KT > gpurun_out/s21_timeline.txt 2>&1
python tools/rocpd_gaps.py $KT > gpurun_out/s21_gaps.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_kt gpurun_out/prof_ks
tail -4 gpurun_out/s21_gpu_tests.txt; head -c 300 gpurun_out/s21_bench_default.json
