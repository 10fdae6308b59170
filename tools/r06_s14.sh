#!/bin/bash
# Round 6, GPU session 14: the batched weight preparation with 16-byte loads / 8-byte stores on interior tiles -- tests, its time in the bench record.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -k "prep or weight or model_fp32 or model_bf16 or state_dict" 2>&1 | tail -3 > gpurun_out/r06_s14_tests.txt
timeout 400 python bench.py --no-cpu-baseline --no-fidelity --steps 30 --warmup 10 > gpurun_out/r06_s14_bench.json 2> gpurun_out/r06_s14_bench.err
cat gpurun_out/r06_s14_tests.txt
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_s14_bench.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["hbm_kernels"]["weight_prep"], r["roofline"]["consistency"])
PY
