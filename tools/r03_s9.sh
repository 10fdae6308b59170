#!/bin/bash
# Round 3, GPU session 9: pooling kernels with every candidate load issued up front (pool_windows.h); the fused stem backward
# tail (VIRTEX_AMD_FUSE_STEM_TAIL) again on those kernels; tile rule of the statistics epilogues
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=$R/virtex_amd/lib
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 oldpool:lib=$L/libvirtex_amd_epipre.so newpool newpool_tail:FUSE_STEM_TAIL=1 \
  newpool_st0:sw.stats_tile=0 newpool_tail_st0:FUSE_STEM_TAIL=1,sw.stats_tile=0 > gpurun_out/s9_ab.txt 2> gpurun_out/s9_ab.err
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "pool or stem or bn or batchnorm" > gpurun_out/s9_tests.txt 2>&1
VIRTEX_AMD_FUSE_STEM_TAIL=1 timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu > gpurun_out/s9_tests_tail.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 > gpurun_out/s9_bench.json 2> gpurun_out/s9_bench.err
cat gpurun_out/s9_ab.txt; tail -3 gpurun_out/s9_tests.txt; tail -3 gpurun_out/s9_tests_tail.txt; tail -3 gpurun_out/s9_ab.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s9_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['roofline'].get('hbm_kernels',{}).items(), key=lambda kv:-kv[1].get('ms_per_step',0))[:14]: print(k, v)
PY
