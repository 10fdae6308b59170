#!/bin/bash
# Round 6, GPU session 8: the stem's streaming kernel with the XCD-major strip order -- time and fetched bytes against the plain
# order (the 1.50x PMC / algorithmic ratio of profiles/r06_traffic_ratio.txt), stem tests, step A/B.
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py tests/test_real_shapes.py -q -m gpu -k "stem" 2>&1 | tail -3 > gpurun_out/r06_s8_tests.txt
timeout 200 python tools/stem_probe.py > gpurun_out/r06_s8_stem_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_stem -- python $R/tools/stem_probe.py > $R/gpurun_out/r06_s8_pmc.log 2>&1
cd $R
python tools/pmc_by_grid.py $(find gpurun_out/pmc_stem -name "*.db" | head -1) stem_stream > gpurun_out/r06_s8_stem_fetch.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/pmc_stem -name "*.db" | head -1) stem_stream > gpurun_out/r06_s8_stem_fetch_dump.txt 2>&1
rm -rf gpurun_out/pmc_stem
timeout 400 python tools/ab_step.py --steps 20 --rounds 3 xcd plain:sw.stem_stream=2 > gpurun_out/r06_s8_ab.txt 2>&1
cat gpurun_out/r06_s8_tests.txt gpurun_out/r06_s8_stem_probe.txt gpurun_out/r06_s8_stem_fetch.txt; head -20 gpurun_out/r06_s8_stem_fetch_dump.txt; tail -3 gpurun_out/r06_s8_ab.txt
