#!/bin/bash
# Round 3, GPU session 12: expand1x1 / stem streaming kernels with a counted prefetch (straight-line drain, pinned hand-over),
# attention staging, unrolled finalize reductions -- against the build of session 10 (libvirtex_amd_s10.so)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=$R/virtex_amd/lib
timeout 900 python tools/ab_step.py --rounds 4 --steps 20 s10:lib=$L/libvirtex_amd_s10.so new > gpurun_out/s12_ab.txt 2> gpurun_out/s12_ab.err
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -x -q -m gpu > gpurun_out/s12_tests.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 > gpurun_out/s12_bench.json 2> gpurun_out/s12_bench.err
cat gpurun_out/s12_ab.txt; tail -3 gpurun_out/s12_tests.txt; tail -3 gpurun_out/s12_ab.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s12_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['roofline'].get('hbm_kernels',{}).items(), key=lambda kv:-kv[1].get('ms_per_step',0))[:22]: print(k, v)
PY
