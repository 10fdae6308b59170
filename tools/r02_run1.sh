#!/bin/bash
# Round 2, first GPU session: new GPU tests, BatchNorm-fusion A/B, full default bench line, kernel traces.
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
nproc > gpurun_out/host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/host.txt; free -g | head -2 >> gpurun_out/host.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/gpu_tests.txt
for m in none bwd fwd both; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --bn-fusion $m --steps 30 --warmup 10 2> gpurun_out/ab_$m.err | tail -1 > gpurun_out/ab_$m.json
done
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 6 --warmup 3 > $R/gpurun_out/prof_ks.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 6 --warmup 3 > $R/gpurun_out/prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 70 > gpurun_out/kernel_stats_serial.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_kt -name "*.db" | head -1) 70 > gpurun_out/kernel_stats.txt
find gpurun_out -name "*.db" -delete
cat gpurun_out/gpu_tests.txt
for m in none bwd fwd both; do cat gpurun_out/ab_$m.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$m', r['ms_per_step'], r['value'])"; done
head -c 600 gpurun_out/bench_default.json
