#!/bin/bash
# Round 6, GPU session 9: the fidelity suite under both strip orders of the stem kernel (the XCD-major order changes the summation
# order of the stem's fp32 BatchNorm partials: how far do the calibrated statistics move?), then the whole GPU suite.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fid_xcd gpurun_out/fid_plain
timeout 900 python -m pytest tests/test_fidelity.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_s9_fidelity_xcd.txt
cp gpurun_out/fidelity_*.json gpurun_out/fid_xcd/
VIRTEX_AMD_STEM_STREAM=2 timeout 900 python -m pytest tests/test_fidelity.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_s9_fidelity_plain.txt
cp gpurun_out/fidelity_*.json gpurun_out/fid_plain/
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/gpu_tests.txt
cat gpurun_out/r06_s9_fidelity_xcd.txt gpurun_out/r06_s9_fidelity_plain.txt gpurun_out/gpu_tests.txt
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/fid_xcd/fidelity_bf16_b*_random*.json")):
    n = os.path.basename(f)
    for tag in ("fid_xcd", "fid_plain"):
        d = json.load(open(f"gpurun_out/{tag}/{n}"))
        k = "hip_bf16_vs_hip_fp32" if "hip_bf16_vs_hip_fp32" in d else "hip_bf16_vs_fp32_oracle"
        o, c = d[k]["backbone"], d["autocast_bf16_vs_fp32_oracle"]["backbone"]
        print(n, tag, "ours median/min/p10", o["median_rel"], o["min_cos"], o["p10_cos"], "| autocast", c["median_rel"], c["min_cos"], c["p10_cos"])
PY
