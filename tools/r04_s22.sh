#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 300 rocprofv3 --pmc $c -d $R/gpurun_out/pj_$c -- python tools/pmc_join_probe.py > $R/gpurun_out/pj_$c.log 2>&1)
  db=$(find $R/gpurun_out/pj_$c -name "*.db" | head -1)
  python $R/tools/pmc_by_grid.py $db "contraction" > $R/gpurun_out/r04_s22_$c.txt 2>&1
done
find $R/gpurun_out -name "*.db" -delete; rm -rf $R/gpurun_out/pj_FETCH_SIZE $R/gpurun_out/pj_WRITE_SIZE
cat $R/gpurun_out/pj_FETCH_SIZE.log | grep -v amdgpu; cat $R/gpurun_out/r04_s22_FETCH_SIZE.txt $R/gpurun_out/r04_s22_WRITE_SIZE.txt
