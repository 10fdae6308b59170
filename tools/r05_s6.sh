#!/bin/bash
# Round 5, GPU session 6: new GPU tests (JPEG decode, replay through RCCL, conv3_bwd after the lean-math change), bench line, step A/B
set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_jpeg.py tests/test_distributed_gpu.py tests/test_replay.py tests/test_kernels.py tests/test_model_parity.py -x -q -m gpu -k "jpeg or coco or rccl or replay or conv3_backward_in_one or fused_conv3_backward or model_bf16_gpu or two_ranks_one_gpu" 2>&1 | tail -8 > gpurun_out/r05_s6_tests.txt
timeout 200 python tools/bench_conv3_bwd.py > gpurun_out/r05_s6_conv3_bwd.txt 2>&1
timeout 300 python tools/ab_step.py --steps 20 --rounds 3 fused off:FUSE_CONV3_BWD=0 > gpurun_out/r05_s6_ab.txt 2>&1
timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r05_s6_bench.json 2> gpurun_out/r05_s6_bench.err
cat gpurun_out/r05_s6_tests.txt gpurun_out/r05_s6_conv3_bwd.txt gpurun_out/r05_s6_ab.txt; tail -c 300 gpurun_out/r05_s6_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_s6_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['config'].get('eager_ms_per_step'), d['config'].get('launch'), d.get('fidelity'))
PY
