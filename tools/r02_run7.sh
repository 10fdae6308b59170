#!/bin/bash
# A/B of the host order of the two heads + full-step timeline dump
set -x
R=$GRAFT_REPO_ROOT; cd $R
rm -f gpurun_out/ab7.txt
for v in "VIRTEX_AMD_HEAD_ORDER_BRANCH_FIRST=0" "X=1" "VIRTEX_AMD_HEAD_ORDER_BRANCH_FIRST=0" "X=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab7.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'])" >> gpurun_out/ab7.txt
done
timeout 600 python -m pytest tests/test_model_parity.py tests/test_distributed_gpu.py -q -m gpu 2>&1 | tail -3 > gpurun_out/gpu_tests7.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_kt7 -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 6 --warmup 3 > $R/gpurun_out/prof_kt7.log 2>&1
cd $R
python tools/rocpd_timeline.py $(find gpurun_out/prof_kt7 -name "*.db" | head -1) 0 40 > gpurun_out/timeline_full.txt 2>&1
find gpurun_out -name "*.db" -delete
cat gpurun_out/ab7.txt gpurun_out/gpu_tests7.txt; head -16 gpurun_out/timeline_full.txt
