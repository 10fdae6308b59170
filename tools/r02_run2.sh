#!/bin/bash
# persistent contraction kernel: correctness on the GPU, per-layer table and step A/B against one tile per block
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_kernels.py tests/test_real_shapes.py tests/test_model_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/gpu_tests2.txt
python tools/bench_layers.py > gpurun_out/layers_persistent.txt 2>&1
VTX_GRID_CAP=1000000 python tools/bench_layers.py > gpurun_out/layers_onetile.txt 2>&1
for cap in 0 1000000; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --grid-cap $cap --steps 30 --warmup 10 2> gpurun_out/cap_$cap.err | tail -1 > gpurun_out/cap_$cap.json
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-fidelity --serial-streams --grid-cap $cap --steps 30 --warmup 10 2>> gpurun_out/cap_$cap.err | tail -1 > gpurun_out/cap_serial_$cap.json
done
cat gpurun_out/gpu_tests2.txt
for f in cap_0 cap_1000000 cap_serial_0 cap_serial_1000000; do python -c "import sys,json; r=json.loads(open('gpurun_out/$f.json').read()); print('$f', r['ms_per_step'], r['value'])"; done
paste <(cut -c1-34,40-60 gpurun_out/layers_persistent.txt) <(cut -c40-60 gpurun_out/layers_onetile.txt) | head -40
