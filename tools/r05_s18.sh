#!/bin/bash
# Round 5, GPU session 18: stock PyTorch-ROCm baseline of BASELINE configs 4 / 5 (the reference's graph on this GPU)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
C4="--textual transdec_postnorm::L4_H1024_A16_F4096 --batch 128"
C5="--visual torchvision::resnet101 --textual transdec_postnorm::L1_H2048_A32_F8192 --batch 64"
timeout 420 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --stock-pytorch-baseline --steps 20 --warmup 10 $C4 > gpurun_out/bench_stock_pytorch_baseline_config4.json 2> gpurun_out/bench_stock_pytorch_baseline_config4.err
timeout 420 python bench.py --no-cpu-baseline --no-fidelity --no-roofline --stock-pytorch-baseline --steps 20 --warmup 10 $C5 > gpurun_out/bench_stock_pytorch_baseline_config5.json 2> gpurun_out/bench_stock_pytorch_baseline_config5.err
for c in 4 5; do python - <<PY
import json
d=json.load(open("gpurun_out/bench_stock_pytorch_baseline_config$c.json"))
print("config $c: ours", d["value"], d["ms_per_step"], "stock", json.dumps(d.get("stock_pytorch_baseline"))[:600])
PY
done
