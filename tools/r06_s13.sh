#!/bin/bash
# Round 6, GPU session 13: the batch-size sweep continued (768, 1024 images per GPU): where does images/sec saturate, what does the envelope refuse?
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for B in 768 1024; do
  timeout 900 python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 20 --warmup 6 > gpurun_out/r06_s13_bench_b$B.json 2> gpurun_out/r06_s13_bench_b$B.err
  tail -c 400 gpurun_out/r06_s13_bench_b$B.err
done
python - <<'PY'
import json
for B in (768, 1024):
    try:
        r = json.loads([l for l in open(f"gpurun_out/r06_s13_bench_b{B}.json") if l.startswith("{")][-1])
        print(B, r["value"], r["ms_per_step"], r["config"]["launch"], r["config"]["peak_memory_gb"], r.get("fidelity", {}).get("backbone"))
    except Exception as e:
        print(B, "no record:", e)
PY
