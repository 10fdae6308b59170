#!/bin/bash
# Round 6, GPU session 12: the bench record's data_parallel fields through RCCL with one forced rank (the only RCCL run a 1-GPU box allows).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VIRTEX_AMD_FORCE_DIST=nccl timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/r06_s12_bench_rccl_one_rank.json 2> gpurun_out/r06_s12_bench.err
tail -c 600 gpurun_out/r06_s12_bench.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_s12_bench_rccl_one_rank.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["config"]["launch"], r.get("data_parallel"))
PY
