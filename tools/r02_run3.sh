#!/bin/bash
# fused tied CE + stem tail + BN apply unroll: GPU tests, A/B, profiles
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/gpu_tests3.txt
python tools/bench_bn_apply.py > gpurun_out/bn_apply_unroll.txt 2>&1
for v in "VIRTEX_AMD_FUSED_CE=0" "VIRTEX_AMD_FUSE_STEM_TAIL=0" "X=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 30 --warmup 10 2> gpurun_out/ab3.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab3.txt
done
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ks -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --serial-streams --steps 6 --warmup 3 > $R/gpurun_out/prof_ks.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 6 --warmup 3 > $R/gpurun_out/prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_ks -name "*.db" | head -1) 70 > gpurun_out/kernel_stats_serial.txt
python tools/rocpd_stats.py $(find gpurun_out/prof_kt -name "*.db" | head -1) 70 > gpurun_out/kernel_stats.txt
python tools/rocpd_gaps.py $(find gpurun_out/prof_kt -name "*.db" | head -1) > gpurun_out/gaps.txt 2>&1
find gpurun_out -name "*.db" -delete
cat gpurun_out/gpu_tests3.txt gpurun_out/ab3.txt gpurun_out/bn_apply_unroll.txt
head -c 400 gpurun_out/bench_default.json
