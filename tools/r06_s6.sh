#!/bin/bash
# Round 6, GPU session 6: the k-major 1x1 weight-gradient class per shape and per library switch, its wave-time split (VERDICT
# item 4c), and the standing robustness evidence on the final build (run-to-run determinism, a sustained 1500-step run).
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 400 python tools/wgrad_shapes_probe.py > gpurun_out/r06_s6_wgrad_shapes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PROBE_PLAIN=1 timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_wgrad -- python $R/tools/wgrad_shapes_probe.py > $R/gpurun_out/r06_s6_pmc.log 2>&1
cd $R
python tools/pmc_by_grid.py $(find gpurun_out/pmc_wgrad -name "*.db" | head -1) contraction > gpurun_out/r06_s6_wgrad_wave_time.txt 2>&1
rm -rf gpurun_out/pmc_wgrad
timeout 400 python tools/determinism.py > gpurun_out/r06_s6_determinism.txt 2>&1
timeout 600 python tools/sustained.py 1500 > gpurun_out/r06_s6_sustained.txt 2>&1
cat gpurun_out/r06_s6_wgrad_shapes.txt; head -60 gpurun_out/r06_s6_wgrad_wave_time.txt; tail -8 gpurun_out/r06_s6_determinism.txt; tail -3 gpurun_out/r06_s6_sustained.txt
