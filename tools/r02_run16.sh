#!/bin/bash
# PMC traffic passes (separate, no trace domains) + A/B of the prologue / pooling changes
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/gpu_tests16.txt
cat gpurun_out/gpu_tests16.txt
rm -f gpurun_out/ab16.txt
PREV=$R/virtex_amd/lib/libvirtex_amd_prev4.so
for rep in 1 2; do
for v in "VIRTEX_AMD_LIB=$PREV" "X=1" "VIRTEX_AMD_BN_UNROLL=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 40 --warmup 10 2> gpurun_out/ab16.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('${v##*/}', r['ms_per_step'], r['value'])" >> gpurun_out/ab16.txt
done; done
cat gpurun_out/ab16.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
cd $R
python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) contraction > gpurun_out/pmc_fetch.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) contraction > gpurun_out/pmc_write.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) bn_ > gpurun_out/pmc_fetch_bn.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) bn_ > gpurun_out/pmc_write_bn.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
grep -A1 "PlainMC\|ConvWgradB" gpurun_out/pmc_fetch.txt | cut -c1-170
