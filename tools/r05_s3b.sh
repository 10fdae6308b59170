#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_join -- python $R/tools/join_shapes_probe.py > $R/gpurun_out/r05_s3_pmc.log 2>&1
cd $R
python tools/pmc_by_grid.py $(find gpurun_out/pmc_join -name "*.db" | head -1) contraction > gpurun_out/r05_s3_join_wave_time.txt 2>&1
rm -rf gpurun_out/pmc_join
cat gpurun_out/r05_s3_join_wave_time.txt
