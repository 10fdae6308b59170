#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 5 --warmup 3 > $R/gpurun_out/prof_tl.log 2>&1
cd $R; db=$(find gpurun_out/prof_tl -name "*.db" | head -1); ls -la $db
python tools/rocpd_timeline.py $db > gpurun_out/timeline.txt 2>&1
cp $db gpurun_out/timeline.db; rm -rf gpurun_out/prof_tl
cat gpurun_out/timeline.txt
