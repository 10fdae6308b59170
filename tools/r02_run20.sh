#!/bin/bash
# PMC traffic passes of the final build (separate passes, no trace domains)
set -x
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- python $R/bench.py --no-cpu-baseline --no-roofline --no-fidelity --steps 2 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
cd $R
python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) contraction > gpurun_out/pmc_fetch.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) contraction > gpurun_out/pmc_write.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_fetch -name "*.db" | head -1) bn_ > gpurun_out/pmc_fetch_bn.txt 2>&1
python tools/pmc_dump.py $(find gpurun_out/prof_write -name "*.db" | head -1) bn_ > gpurun_out/pmc_write_bn.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
cat gpurun_out/pmc_fetch_bn.txt | head -30
