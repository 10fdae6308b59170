#!/bin/bash
# Round 5, GPU session 3: where the tiled join kernel loses time per stage shape (wave-time split by grid), and the new bench fields.
set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 200 python tools/join_shapes_probe.py > gpurun_out/r05_s3_join_shapes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_join -- python $R/tools/join_shapes_probe.py > $R/gpurun_out/r05_s3_pmc.log 2>&1
cd $R
python tools/pmc_by_grid.py $(find gpurun_out/pmc_join -name "*.db" | head -1) contraction > gpurun_out/r05_s3_join_wave_time.txt 2>&1
rm -rf gpurun_out/pmc_join
timeout 400 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-fidelity > gpurun_out/r05_s3_bench.json 2> gpurun_out/r05_s3_bench.err
cat gpurun_out/r05_s3_join_shapes.txt gpurun_out/r05_s3_join_wave_time.txt
tail -c 400 gpurun_out/r05_s3_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_s3_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['config'].get('eager_ms_per_step'), d['config'].get('launch'), d['roofline'].get('dominant_contraction'))
PY
