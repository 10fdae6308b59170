#!/bin/bash
# Round 4, GPU session 18: launch replay (virtex_amd/replay.py) -- GPU tests, BASELINE configs 2 / 4 / 5 eager vs replay
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_replay.py tests/test_model_parity.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r04_s18_tests.txt
cat gpurun_out/r04_s18_tests.txt
Q="--no-cpu-baseline --no-fidelity --no-roofline --steps 30 --warmup 8"
for g in eager replay; do
  timeout 600 python bench.py $Q --launch $g > gpurun_out/r04_s18_cfg2_$g.json 2> gpurun_out/r04_s18_cfg2_$g.err
  timeout 600 python bench.py $Q --launch $g --textual transdec_postnorm::L4_H1024_A16_F4096 --batch 128 > gpurun_out/r04_s18_cfg4_$g.json 2> gpurun_out/r04_s18_cfg4_$g.err
  timeout 600 python bench.py $Q --launch $g --visual torchvision::resnet101 --textual transdec_postnorm::L1_H2048_A32_F8192 --batch 64 > gpurun_out/r04_s18_cfg5_$g.json 2> gpurun_out/r04_s18_cfg5_$g.err
done
for f in gpurun_out/r04_s18_cfg*.json; do echo $f; python -c "
import json,sys
try:
    r=json.loads(open('$f').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['config'].get('launch'), r['config'].get('final_loss'))
except Exception as e: print('no json', e)
"; tail -2 ${f%.json}.err | grep -v amdgpu.ids; done
