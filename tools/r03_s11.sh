#!/bin/bash
# Round 3, GPU session 11: LayerNorm / embedding / attention kernels with their loads issued up front, against the build of session 10
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=$R/virtex_amd/lib
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 s10:lib=$L/libvirtex_amd_s10.so new > gpurun_out/s11_ab.txt 2> gpurun_out/s11_ab.err
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "attention or attn or layernorm or embed" > gpurun_out/s11_tests.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 > gpurun_out/s11_bench.json 2> gpurun_out/s11_bench.err
cat gpurun_out/s11_ab.txt; tail -3 gpurun_out/s11_tests.txt; tail -3 gpurun_out/s11_ab.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s11_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['roofline'].get('hbm_kernels',{}).items(), key=lambda kv:-kv[1].get('ms_per_step',0))[:40]: print(k, v)
PY
