#!/bin/bash
# Round 3, GPU session 17: evidence of the build at HEAD -- GPU suite (complete), default bench line, kernel traces (three
# streams: timeline + gaps; serial streams: per-kernel stats), PMC traffic passes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/s17_gpu_tests.txt
python bench.py > gpurun_out/s17_bench_default.json 2> gpurun_out/s17_bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/s17_prof_kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 9 --warmup 3 > $R/gpurun_out/s17_prof_ks.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/s17_prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/s17_prof_write.log 2>&1
cd $R
KT=$(find gpurun_out/prof_kt -name "*.db" | head -1); KS=$(find gpurun_out/prof_ks -name "*.db" | head -1)
python tools/rocpd_stats.py $KT 80 > gpurun_out/s17_kernel_stats.txt
python tools/rocpd_stats.py $KS 80 > gpurun_out/s17_kernel_stats_serial.txt
python tools/rocpd_timeline.py $KT > gpurun_out/s17_timeline.txt 2>&1
python tools/rocpd_gaps.py $KT > gpurun_out/s17_gaps.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) "" > gpurun_out/s17_pmc_fetch.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) "" > gpurun_out/s17_pmc_write.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_kt gpurun_out/prof_ks gpurun_out/prof_fetch gpurun_out/prof_write
tail -8 gpurun_out/s17_gpu_tests.txt; head -c 600 gpurun_out/s17_bench_default.json
