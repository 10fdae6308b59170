#!/bin/bash
# Round 3, GPU session 5: GPU suite on the narrow-finalize build, A/B of the split weight preparation, tile sweep of the weight
# gradients on the repaired pipeline, the last 2 ms of the step's timeline
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/s5_gpu_tests.txt
python tools/ab_step.py --rounds 3 --steps 20 joint_prep:models.SPLIT_WEIGHT_PREP=0 split_prep:models.SPLIT_WEIGHT_PREP=1 > gpurun_out/s5_ab.txt 2> gpurun_out/s5_ab.err
python tools/sweep_tiles.py -1,0,1,2,3,4,5 wgrad > gpurun_out/s5_sweep_wgrad.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --no-cpu-baseline --no-fidelity --no-roofline --steps 9 --warmup 3 > $R/gpurun_out/s5_prof_kt.log 2>&1
cd $R
KT=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/rocpd_timeline.py $KT 23.5 27.0 > gpurun_out/s5_timeline_tail.txt 2>&1
python tools/rocpd_timeline.py $KT 0.0 1.5 > gpurun_out/s5_timeline_head.txt 2>&1
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof_kt
python bench.py --no-cpu-baseline --no-fidelity --steps 30 --warmup 10 > gpurun_out/s5_bench.json 2> gpurun_out/s5_bench.err
tail -6 gpurun_out/s5_gpu_tests.txt; cat gpurun_out/s5_ab.txt; head -c 300 gpurun_out/s5_bench.json
