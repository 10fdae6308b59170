#!/bin/bash
# Round 5, GPU session 16: GPU tests of the pre-norm head; every dispatch of the text part of one replayed step (all queues) --
# why the compute queue waits 1.7 ms for the second caption head
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "prenorm or dropout_bwd or incremental or rccl" > gpurun_out/r05_s16_tests.txt 2>&1
tail -3 gpurun_out/r05_s16_tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/r05_s16_prof_kt.log 2>&1
cd $R
KT=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/rocpd_timeline.py $KT 7.8 13.4 > gpurun_out/r05_s16_timeline_text.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_kt
head -30 gpurun_out/r05_s16_timeline_text.txt
