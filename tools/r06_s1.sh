#!/bin/bash
# Round 6, GPU session 1: the new parity tests (single Bottleneck at a flat 1e-3, bf16 of BASELINE configs 4 / 5 against the
# oracle), the whole GPU suite after the ResNet schedule was split into block functions, and a default bench line with the
# new roofline headline (largest single kernel instantiation), the consistency verdict and the optimizer's real byte count.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r06_s1_tests.txt
timeout 600 python bench.py > gpurun_out/r06_s1_bench.json 2> gpurun_out/r06_s1_bench.err
cat gpurun_out/r06_s1_tests.txt
tail -c 1500 gpurun_out/r06_s1_bench.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_s1_bench.json") if l.startswith("{")][-1])
roof = r["roofline"]
print(r["value"], r["ms_per_step"], roof["kernel"], roof["frac"], roof["bound"], roof["consistency"], roof["dominant_family"])
print({k: v for k, v in roof["hbm_kernels"].items() if k.startswith("optimizer")})
PY
