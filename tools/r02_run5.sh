#!/bin/bash
# round-2 evidence run: GPU tests, A/B of the shared projection, default bench, kernel traces, PMC traffic
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/gpu_tests5.txt
rm -f gpurun_out/ab5.txt
for v in "VIRTEX_AMD_SHARE_VISUAL_PROJECTION=0" "X=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab5.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab5.txt
done
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 6 --warmup 3 > $R/gpurun_out/prof_ks.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 6 --warmup 3 > $R/gpurun_out/prof_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 2 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 80 > gpurun_out/kernel_stats_serial.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_kt -name "*.db" | head -1) 80 > gpurun_out/kernel_stats.txt
python tools/rocpd_timeline.py $(find gpurun_out/prof_kt -name "*.db" | head -1) > gpurun_out/timeline.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) bn_ > gpurun_out/pmc_fetch_bn.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) bn_ > gpurun_out/pmc_write_bn.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) contraction > gpurun_out/pmc_fetch.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) contraction > gpurun_out/pmc_write.txt 2>&1
find gpurun_out -name "*.db" -delete
cat gpurun_out/gpu_tests5.txt gpurun_out/ab5.txt
head -c 300 gpurun_out/bench_default.json
