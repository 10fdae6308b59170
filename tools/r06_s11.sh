#!/bin/bash
# Round 6, GPU session 11: the stem's pooling tails in XCD-major block order -- time and fetched bytes against the plain order.
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "maxpool or stem or pool" 2>&1 | tail -3 > gpurun_out/r06_s11_tests.txt
timeout 200 python tools/pool_probe.py > gpurun_out/r06_s11_pool_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for O in 1 0; do
  POOL_XCD=$O timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_pool$O -- python $R/tools/pool_probe.py > $R/gpurun_out/r06_s11_pmc$O.log 2>&1
  python $R/tools/pmc_dump.py $(find $R/gpurun_out/pmc_pool$O -name "*.db" | head -1) pool > $R/gpurun_out/r06_s11_fetch_order$O.txt 2>&1
  rm -rf $R/gpurun_out/pmc_pool$O
done
cd $R
timeout 400 python tools/ab_step.py --steps 20 --rounds 3 xcd plain:sw.pool_xcd=0 > gpurun_out/r06_s11_ab.txt 2>&1
cat gpurun_out/r06_s11_tests.txt gpurun_out/r06_s11_pool_probe.txt; grep -A1 "pool\|maxpool" gpurun_out/r06_s11_fetch_order1.txt | head -12; grep -A1 "pool\|maxpool" gpurun_out/r06_s11_fetch_order0.txt | head -12; tail -3 gpurun_out/r06_s11_ab.txt
