#!/bin/bash
# Round 3, GPU session 20: embedding backward with lane-contiguous atomics (prev = build before), stand-alone BatchNorm
# reductions with interleaved blocks (bn_reduce_form 1: another fp32 summation order -> the gradient-bound tests with it)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=$R/virtex_amd/lib
timeout 900 python tools/ab_step.py --rounds 3 --steps 20 prev:lib=$L/libvirtex_amd_prev.so new new_inter:sw.bn_reduce_form=1 > gpurun_out/s20_ab.txt 2> gpurun_out/s20_ab.err
timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 > gpurun_out/s20_bench.json 2> gpurun_out/s20_bench.err
VIRTEX_AMD_BN_REDUCE_FORM=1 timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 > gpurun_out/s20_bench_inter.json 2> gpurun_out/s20_bench_inter.err
VIRTEX_AMD_BN_REDUCE_FORM=1 timeout 1200 python -m pytest tests/test_fidelity.py tests/test_model_parity.py tests/test_kernels.py -x -q -m gpu > gpurun_out/s20_tests_inter.txt 2>&1
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "embed" > gpurun_out/s20_tests.txt 2>&1
cat gpurun_out/s20_ab.txt; tail -3 gpurun_out/s20_tests_inter.txt; tail -2 gpurun_out/s20_tests.txt; tail -3 gpurun_out/s20_ab.err
python - <<'PY'
import json
for f in ['gpurun_out/s20_bench.json','gpurun_out/s20_bench_inter.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    h=d['roofline']['hbm_kernels']
    print(f, d['ms_per_step'], {k:h[k] for k in ('embedding_bwd','bn_bwd_reduce','bn_fwd_reduce') if k in h})
PY
