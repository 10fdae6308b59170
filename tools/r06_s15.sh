#!/bin/bash
# Round 6, GPU session 15: the optimizer kernel with 16-byte accesses -- optimizer / replay tests, its time in the bench record.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_optimizer.py tests/test_replay.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_s15_tests.txt
timeout 400 python bench.py --no-cpu-baseline --no-fidelity --steps 30 --warmup 10 > gpurun_out/r06_s15_bench.json 2> gpurun_out/r06_s15_bench.err
cat gpurun_out/r06_s15_tests.txt
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_s15_bench.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["hbm_kernels"]["optimizer_step"], r["roofline"]["hbm_kernels"]["weight_prep"], r["roofline"]["consistency"])
PY
