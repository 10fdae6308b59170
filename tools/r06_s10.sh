#!/bin/bash
# Round 6, GPU session 10: the record at HEAD -- the whole GPU suite, smoke(), the default bench line (with config.best_batch of this build).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "rc=$?" >> gpurun_out/smoke.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/gpu_tests.txt; tail -3 gpurun_out/smoke.txt
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["kernel"][:60], r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline"]["consistency"], r["config"]["best_batch"])
PY
