#!/bin/bash
# Round 6, GPU session 4: split-K reductions batched per Bottleneck / per decoder layer (ops.splitk_batch) -- tests on hardware
# (bit-identical gradients, launch counts), replay and data-parallel tests with it on, step-level A/B, reduce launches per step.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_model_parity.py tests/test_replay.py tests/test_distributed_gpu.py tests/test_optimizer.py -q -m gpu -k "several_weight_gradients or model or replay or rccl or gradient or step" 2>&1 | tail -6 > gpurun_out/r06_s4_tests.txt
timeout 400 python tools/ab_step.py --steps 20 --rounds 3 batched off:splitk.enabled=0 > gpurun_out/r06_s4_ab.txt 2>&1
timeout 400 python tools/ab_step.py --steps 20 --rounds 2 batched:serial=1 off:splitk.enabled=0,serial=1 > gpurun_out/r06_s4_ab_serial.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-fidelity --steps 20 --warmup 10 > gpurun_out/r06_s4_bench.json 2> gpurun_out/r06_s4_bench.err
cat gpurun_out/r06_s4_tests.txt; tail -3 gpurun_out/r06_s4_ab.txt; tail -3 gpurun_out/r06_s4_ab_serial.txt
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_s4_bench.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["config"]["launch"], {k: v for k, v in r["roofline"]["hbm_kernels"].items() if "splitk" in k})
PY
