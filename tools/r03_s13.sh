#!/bin/bash
# Round 3, GPU session 13: the stem's weight gradient on one 64x256 tile (dy read once per K slice); bs 128 per-kernel rates
# (do the 100-400 MB tensors of bs 128 stream faster out of the 256 MB Infinity Cache?)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python tools/ab_step.py --rounds 4 --steps 20 tile0:sw.tile64x256=0 tile1:sw.tile64x256=1 > gpurun_out/s13_ab.txt 2> gpurun_out/s13_ab.err
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -x -q -m gpu -k "wgrad or stem or conv" > gpurun_out/s13_tests.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 > gpurun_out/s13_bench.json 2> gpurun_out/s13_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 --batch 128 > gpurun_out/s13_bench_b128.json 2> gpurun_out/s13_bench_b128.err
cat gpurun_out/s13_ab.txt; tail -3 gpurun_out/s13_tests.txt; tail -3 gpurun_out/s13_ab.err
python - <<'PY'
import json
for f in ['gpurun_out/s13_bench.json','gpurun_out/s13_bench_b128.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'])
    for k,v in sorted(d['roofline'].get('hbm_kernels',{}).items(), key=lambda kv:-kv[1].get('ms_per_step',0))[:8]: print('  ', k, v)
    for k,v in sorted(d['roofline'].get('mfma_kernels',{}).items(), key=lambda kv:-kv[1].get('ms_per_step',0))[:60]:
        if 'ConvWgradB' in k or 'stem' in k.lower(): print('  ', k[:150], v)
PY
