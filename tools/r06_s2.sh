#!/bin/bash
# Round 6, GPU session 2: bn3's backward folded into conv3's weights (csrc/bn_fold.hip) -- kernel + schedule tests on hardware,
# the fidelity suite with the fold on (does the bf16 step stay inside the autocast band?), step-level A/B fold on / off (three
# streams and serial), the recalibrated single-Bottleneck bf16 test.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -s -m gpu -k "folded or single_bottleneck" 2>&1 | grep -v Warning | tail -12 > gpurun_out/r06_s2_tests.txt
timeout 900 python -m pytest tests/test_fidelity.py tests/test_replay.py -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_s2_fidelity_tests.txt
timeout 400 python tools/ab_step.py --steps 20 --rounds 3 fold off:FUSE_BN3_FOLD=0 > gpurun_out/r06_s2_ab.txt 2>&1
timeout 400 python tools/ab_step.py --steps 20 --rounds 2 fold:serial=1 off:FUSE_BN3_FOLD=0,serial=1 > gpurun_out/r06_s2_ab_serial.txt 2>&1
cat gpurun_out/r06_s2_tests.txt gpurun_out/r06_s2_fidelity_tests.txt
tail -8 gpurun_out/r06_s2_ab.txt; tail -6 gpurun_out/r06_s2_ab_serial.txt
