#!/bin/bash
# HBM traffic per SHAPE of the fused-BatchNorm-backward 1x1 input-gradient class (PMC ratio 1.28 against the algorithmic bytes)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c -d $R/gpurun_out/pg_$c -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --launch eager --steps 3 --warmup 2 > $R/gpurun_out/pg_$c.log 2>&1
  db=$(find $R/gpurun_out/pg_$c -name "*.db" | head -1)
  python $R/tools/pmc_by_grid.py $db "EpiStore<unsigned short, 2>" > $R/gpurun_out/r04_s21_$c.txt 2>&1
  python $R/tools/pmc_by_grid.py $db "EpiStore<unsigned short, 1>" >> $R/gpurun_out/r04_s21_$c.txt 2>&1
done
find $R/gpurun_out -name "*.db" -delete; rm -rf $R/gpurun_out/pg_FETCH_SIZE $R/gpurun_out/pg_WRITE_SIZE
head -80 $R/gpurun_out/r04_s21_FETCH_SIZE.txt
