#!/bin/bash
# Round 4, GPU session 6: the step as one hipGraph -- new GPU tests, BASELINE configs 2 / 4 / 5 with and without the graph
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_optimizer.py tests/test_model_parity.py -q -m gpu -x -k "device_side or dropout_epoch or bf16_gpu" 2>&1 | tail -15 > gpurun_out/r04_s6_tests.txt
Q="--no-cpu-baseline --no-fidelity --no-roofline --steps 30 --warmup 8"
for g in graph eager; do
  timeout 600 python bench.py $Q --launch $g > gpurun_out/r04_s6_cfg2_graph_$g.json 2> gpurun_out/r04_s6_cfg2_graph_$g.err
  timeout 600 python bench.py $Q --launch $g --textual transdec_postnorm::L4_H1024_A16_F4096 --batch 128 > gpurun_out/r04_s6_cfg4_graph_$g.json 2> gpurun_out/r04_s6_cfg4_graph_$g.err
  timeout 600 python bench.py $Q --launch $g --visual torchvision::resnet101 --textual transdec_postnorm::L1_H2048_A32_F8192 --batch 64 > gpurun_out/r04_s6_cfg5_graph_$g.json 2> gpurun_out/r04_s6_cfg5_graph_$g.err
done
cat gpurun_out/r04_s6_tests.txt
for f in gpurun_out/r04_s6_cfg*_graph_*.json; do echo $f; python -c "
import json,sys
try:
    r=json.loads(open('$f').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['config'].get('launch'), r['config'].get('final_loss'))
except Exception as e: print('no json', e)
"; tail -2 ${f%.json}.err; done
